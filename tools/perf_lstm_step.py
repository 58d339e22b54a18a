#!/usr/bin/env python3
"""Isolated timing of the fp32 ConvLSTM recurrent steps of the shipped shape (B=8, 96x96; F=16 and F=2), HIP events over a chain
of dependent steps as in the T=24 train step:  python tools/perf_lstm_step.py [reps]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402

from downscaling.engine.hipops import ConvGeom, HipOps  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ops = HipOps("cuda:0")
    dev = ops.device
    g = ConvGeom(3, 3, 1, 1)
    B, S, T = 8, 96, 24
    for F, cp in ((16, 16), (2, 4)):
        h = torch.randn(T * B, S, S, cp, device=dev)
        gates = torch.randn(T * B, S, S, 4 * F, device=dev)
        c = torch.randn(T * B, S, S, F, device=dev)
        dgates, dh, dc = torch.randn_like(gates), torch.randn_like(h), [torch.randn(B, S, S, F, device=dev) for _ in range(2)]
        pk = ops.pack_weights((torch.randn(3, 3, F, 4 * F, device=dev) * 0.1).contiguous())
        def fwd_chain():
            for t in range(1, T):
                sl, pv = slice(t * B, (t + 1) * B), slice((t - 1) * B, t * B)
                ops.convlstm_step(h[pv], pk, gates[sl], c[pv], c[sl], h[sl], g, F)
        def fwd_unfused():
            for t in range(1, T):
                sl, pv = slice(t * B, (t + 1) * B), slice((t - 1) * B, t * B)
                ops.conv_fwd(h[pv], pk, None, gates[sl], g, act=False, accumulate=True)
                ops.lstm_fwd(gates[sl].view(-1, 4 * F), c[pv].view(-1, F), c[sl].view(-1, F), h[sl].view(-1, cp), F)
        def bwd_chain():
            for t in range(T - 1, 0, -1):
                sl, pv, pp = slice(t * B, (t + 1) * B), slice((t - 1) * B, t * B), slice((t - 2) * B, (t - 1) * B)
                ops.convlstm_bwd_step(dgates[sl], pk, dh[pv], gates[pv], c[pp] if t > 1 else None, c[pv], dc[t & 1], dgates[pv],
                                      dc[(t - 1) & 1] if t > 1 else None, g, F)
        for name, fn in (("fwd fused", fwd_chain), ("fwd conv+cell", fwd_unfused), ("bwd fused", bwd_chain)):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f"F={F:2d} {name:14s}: {e0.elapsed_time(e1) / reps / (T - 1) * 1e3:7.1f} us per timestep (chain of {T - 1})")


if __name__ == "__main__":
    main()
