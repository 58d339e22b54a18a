#!/usr/bin/env python3
"""How well do whole discriminator passes (forward + weights-only backward) of different networks overlap on different
streams?  Three networks of the same graph; the same three passes one after the other on one stream, then side by side on
three measured-concurrent streams.   python tools/probe_passes.py [S=96] [T=24] [B=8] [graphs=1]"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402


def main():
    S, T, B = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 96), (2, 24), (3, 8)))
    if len(sys.argv) > 4 and sys.argv[4] == "0":
        os.environ["WDG_CHAIN_GRAPHS"] = "0"
    from downscaling.engine import runtime
    from downscaling.engine.networks import DiscriminatorNet
    ops = runtime.get_ops()
    dev = ops.device
    nets = [DiscriminatorNet(ops, S, S, 3, 2, T, seed=2 + i) for i in range(3)]
    for n in nets:
        n.wgrad_stream = False
    low = torch.randn(B, T, S, S, 3, device=dev)
    high_tm = torch.randn(T * B, S, S, 4, device=dev)
    high_tm[..., 2:] = 0
    dscore = torch.full((B,), 1.0 / B, device=dev)
    for n in nets:
        n.set_low(low)

    def one(n):
        n.set_high_tm(high_tm, B)
        n.forward(B, training=True)
        n.params.zero_grad()
        n.backward(B, dscore, need_wgrad=True, need_input_grad=False)

    main_s = torch.cuda.current_stream(dev)
    streams = ops.concurrent_streams(3)

    def serial():
        for n in nets:
            one(n)

    def concurrent(k):
        for s in streams[:k]:
            s.wait_stream(main_s)
        for n, s in zip(nets[:k], streams[:k]):
            with torch.cuda.stream(s):
                one(n)
        for s in streams[:k]:
            main_s.wait_stream(s)

    def timed(fn, reps=5):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t1 = timed(lambda: one(nets[0]))
    t3 = timed(serial)
    c2 = timed(lambda: concurrent(2))
    c3 = timed(lambda: concurrent(3))
    print(f"S={S} T={T} B={B} graphs={os.environ.get('WDG_CHAIN_GRAPHS', '1')}: one pass {t1:.2f} ms; three in sequence {t3:.2f} ms; "
          f"two side by side {c2:.2f} ms; three side by side {c3:.2f} ms")


if __name__ == "__main__":
    main()
