#!/usr/bin/env python3
"""The 16-bit generator ConvLSTM recurrence (models.py:45, F = 128, 24 x 24 maps, 16 tiles, T = 24) alone: the 24 step
launches replayed from a HIP graph, HIP-event timed; trailing key=int pairs go to wdg_set_tuning (A/B of tile choices).
    python tools/perf_step16.py [bf16|fp16] [key=int ...]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402

from downscaling.engine.hipops import ConvGeom, HipOps  # noqa: E402


def main():
    fmt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    ops = HipOps("cuda:0")
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        assert ops.lib.wdg_set_tuning(k.encode(), int(v)) == 0, kv
    B, T, S, F = 16, 24, 24, 128
    dev = ops.device
    g = ConvGeom(3, 3, 1, 1)
    w = (torch.randn(3, 3, F, 4 * F, device=dev) * 0.02).contiguous()
    pk = ops.pack_weights(w)
    gates = torch.randn(T * B, S, S, 4 * F, device=dev) * 0.5
    h = torch.zeros(T * B, S, S, F, device=dev)
    c = torch.zeros(T * B, S, S, F, device=dev)
    assert ops.convlstm16_supported(h[:B], gates[:B], pk, g, F)

    def loop():
        for t in range(T):
            sl, pv = slice(t * B, (t + 1) * B), slice((t - 1) * B, t * B)
            ops.convlstm16_step(h[pv] if t else None, pk, gates[sl], c[pv] if t else None, c[sl], h[sl], g, F, fmt=fmt)
    loop()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loop()
    for _ in range(3):
        graph.replay()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{fmt} {sys.argv[2:]}: {1e3 * ts[len(ts) // 2] / T:.1f} us per step (median of 9 replays of {T} steps), checksum {float(h.double().abs().sum()):.6e}")


if __name__ == "__main__":
    main()
