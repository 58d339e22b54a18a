#!/usr/bin/env python3
"""Micro-benchmark of the 16-bit forward convolutions of the shipped generator shape G(96, T=24) on 16 tiles
(conv_patch_h16.hip vs the gather kernel), HIP-event timing, interleaved rounds in ONE process.
    python tools/perf_patch.py [reps] [key=int[,int...] ...]
Every trailing key=v1,v2,... adds columns: the layer is timed with wdg_set_tuning(key, v) for each v (A/B runs; the
patch_dbg values need a library built with -DWDG_PATCH_EXPERIMENTS and give wrong results by design)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402

from downscaling.engine.hipops import ConvGeom, HipOps  # noqa: E402

N = 16 * 24
SHAPES = [
    ("c0 8x8s2 23->128 @96", (N, 96, 96, 23), 128, 8, 2, 3),
    ("c2 4x4s2 128->128 @48", (N, 48, 48, 128), 128, 4, 2, 1),
    ("lstm-x 3x3 128->512 @24", (N, 24, 24, 128), 512, 3, 1, 1),
    ("lstm-h 3x3 128->512 @24 (1 step)", (16, 24, 24, 128), 512, 3, 1, 1),
    ("c5 3x3 128->64 @24", (N, 24, 24, 128), 64, 3, 1, 1),
    # the column GEMM of the column-form upsample + transposed 5x5 layer: the TRANSPOSED direction of a 1x1 plan (400 -> 160)
    ("c9 col GEMM 160->400 @48 (T)", (N, 48, 48, 400), 160, 1, 1, 0),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    cols = [("default", None, 0)]
    for kv in sys.argv[2:]:
        k, vs = kv.split("=")
        cols += [(f"{k}={v}", k.encode(), int(v)) for v in vs.split(",")]
    ops = HipOps("cuda:0")
    lib = ops.lib
    cases = []
    for name, (n, H, W, cin), cout, k, s, p in SHAPES:
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        x = torch.randn(n, H, W, (cin + 3) // 4 * 4, device=ops.device)
        if cin % 4:
            x[..., cin:] = 0
        y = torch.empty(n, Ho, Wo, cout, device=ops.device)
        pk = ops.pack_weights((torch.randn(k, k, cin, cout, device=ops.device) * 0.05).contiguous())
        b = torch.randn(cout, device=ops.device)
        g = ConvGeom(k, k, s, p)
        if name.endswith("(T)"):
            fn = lambda x=x, pk=pk, y=y, g=g: ops.conv_dgrad_bf16(y, pk, x, g)
        else:
            fn = lambda x=x, pk=pk, y=y, g=g, b=b: ops.conv_fwd_bf16(x, pk, b, y, g, act=True)
        cases.append((name, fn, 2.0 * n * Ho * Wo * cout * k * k * cin, 4.0 * (x.numel() + y.numel())))
    times = {(c[0], col[0]): [] for c in cases for col in cols}
    for r in range(reps + 1):
        for label, key, val in cols:
            if key is not None:
                assert lib.wdg_set_tuning(key, val) == 0, label
            for name, fn, flops, nbytes in cases:
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    times[(name, label)].append(e0.elapsed_time(e1))
            if key is not None:
                lib.wdg_set_tuning(key, 1 if key in (b"patch_h16", b"patch_nloop", b"patch_flat") else 0)   # back to the default
    print(f"{'layer':36s} " + " ".join(f"{c[0]:>16s}" for c in cols) + "   (us; default: TFLOP/s, in+out TB/s)")
    for name, fn, flops, nbytes in cases:
        med = lambda l: sorted(l)[len(l) // 2]
        row = f"{name:36s} " + " ".join(f"{med(times[(name, c[0])]) * 1e3:16.1f}" for c in cols)
        t0 = med(times[(name, 'default')])
        print(row + f"   {flops / t0 * 1e-9:6.1f} {nbytes / t0 * 1e-9:5.2f}")


if __name__ == "__main__":
    main()
