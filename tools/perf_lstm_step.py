#!/usr/bin/env python3
"""Isolated timing of the fp32 ConvLSTM recurrent steps of the shipped shape (B=8, 96x96; F=16 and F=2), HIP events over a chain
of dependent steps as in the T=24 train step:  python tools/perf_lstm_step.py [reps [batch]]"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402

from downscaling.engine.hipops import ConvGeom, HipOps  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    ops = HipOps("cuda:0")
    dev = ops.device
    g = ConvGeom(3, 3, 1, 1)
    B, S, T = batch, 96, 24
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
            print(f"F={F:2d} {name:14s}: {e0.elapsed_time(e1) / reps / (T - 1) * 1e3:7.1f} us per timestep (chain of {T - 1}, batch {B})")
        if F == 16 and not os.environ.get("WDG_PERF_NO2"):
            # two independent chains on two streams (the train step's real and gradient-penalty passes): do they share the chip?
            h2, gates2, c2 = torch.randn_like(h), torch.randn_like(gates), torch.randn_like(c)
            def fwd_chain2():
                for t in range(1, T):
                    sl, pv = slice(t * B, (t + 1) * B), slice((t - 1) * B, t * B)
                    ops.convlstm_step(h2[pv], pk, gates2[sl], c2[pv], c2[sl], h2[sl], g, F)
            graphs = []
            for fn in (fwd_chain, fwd_chain2):
                fn()
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    fn()
                graphs.append(gr)
            s2 = torch.cuda.Stream()
            for mode in ("one graph", "two graphs, one stream", "two graphs, two streams"):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    graphs[0].replay()
                    if mode == "two graphs, one stream":
                        graphs[1].replay()
                    elif mode == "two graphs, two streams":
                        s2.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(s2):
                            graphs[1].replay()
                        torch.cuda.current_stream().wait_stream(s2)
                e1.record()
                torch.cuda.synchronize()
                print(f"F=16 fwd {mode:26s}: {e0.elapsed_time(e1) / reps / (T - 1) * 1e3:7.1f} us per timestep of one chain")
    generator_step(ops, reps, B)


def generator_step(ops, reps, B, T=24):
    """The generator's 128-feature recurrent step on the 24 x 24 map of the shipped 96 x 96 tiles (models.py:45): GEMM form, cell on the accumulators."""
    dev, F, S = ops.device, 128, 24
    g = ConvGeom(3, 3, 1, 1)
    h = torch.randn(T * B, S, S, F, device=dev) * 0.1
    gates = torch.randn(T * B, S, S, 4 * F, device=dev)
    c = torch.randn(T * B, S, S, F, device=dev)
    pk = ops.pack_weights((torch.randn(3, 3, F, 4 * F, device=dev) * 0.02).contiguous())

    def chain():
        for t in range(1, T):
            sl, pv = slice(t * B, (t + 1) * B), slice((t - 1) * B, t * B)
            ops.convlstm_step(h[pv], pk, gates[sl], c[pv], c[sl], h[sl], g, F)
    chain()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        chain()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps / (T - 1) * 1e3
    fl = 2.0 * B * S * S * 9 * F * 4 * F
    print(f"F=128 generator step (24x24, batch {B}): {us:7.1f} us per timestep = {fl / us * 1e-6:6.1f} TFLOP/s")


if __name__ == "__main__":
    main()
