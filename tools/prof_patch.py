#!/usr/bin/env python3
"""Per-phase shader-clock totals of conv_patch_h16.hip, summed over all workgroups of a launch, from a measurement build
(-DWDG_PATCH_PROF=1, passed as WDG_LIB):  WDG_LIB=.../libwdgan_pp.so python tools/prof_patch.py
Phases (wave 0 of every workgroup, s_memtime): 0 index arithmetic + K-step tables, 1 chunk barrier + patch (requests, conversion,
LDS stores), 2 first weight stages + barrier, 3 the stage loop, 4 epilogue until its stores are issued, 5 until they completed."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402

from downscaling.engine.hipops import ConvGeom, HipOps  # noqa: E402

N = 16 * 24
SHAPES = [
    ("c0 8x8s2 23->128 @96 (out fp32)", (N, 96, 96, 23), 128, 8, 2, 3),
    ("c2 4x4s2 128->128 @48", (N, 48, 48, 128), 128, 4, 2, 1),
    ("lstm-x 3x3 128->512 @24", (N, 24, 24, 128), 512, 3, 1, 1),
    ("c5 3x3 128->64 @24", (N, 24, 24, 128), 64, 3, 1, 1),
]
NAMES = ["tables", "patch load", "first stages", "stage loop", "epilogue issue", "stores complete"]


def main():
    ops = HipOps("cuda:0")
    lib = ops.lib
    lib.wdg_patch_prof.restype = C.c_int
    lib.wdg_patch_prof.argtypes = [C.c_void_p, C.c_int]
    out = (C.c_ulonglong * 8)()
    for name, (n, H, W, cin), cout, k, s, p in SHAPES:
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        x = torch.randn(n, H, W, (cin + 3) // 4 * 4, device=ops.device)
        if cin % 4:
            x[..., cin:] = 0
        y = torch.empty(n, Ho, Wo, cout, device=ops.device)
        pk = ops.pack_weights((torch.randn(k, k, cin, cout, device=ops.device) * 0.05).contiguous())
        b = torch.randn(cout, device=ops.device)
        g = ConvGeom(k, k, s, p)
        fn = lambda: ops.conv_fwd_bf16(x, pk, b, y, g, act=True)      # noqa: E731
        fn()
        torch.cuda.synchronize()
        assert lib.wdg_patch_prof(None, 1) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        assert lib.wdg_patch_prof(out, 0) == 0
        nwg = out[7]
        tot = sum(out[:6])
        print(f"{name}: {e0.elapsed_time(e1) * 1e3:.1f} us, {nwg} workgroups, {tot / max(nwg, 1):.0f} clocks per workgroup (wave 0)")
        for nm, v in zip(NAMES, out[:6]):
            print(f"    {nm:18s} {v / max(nwg, 1):10.0f} clocks  {v / max(tot, 1):6.1%}")


if __name__ == "__main__":
    main()
