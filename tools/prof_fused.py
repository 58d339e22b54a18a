#!/usr/bin/env python3
"""Per-phase shader-clock totals of the fused 16-bit upsample + transposed-conv kernel (workgroup 0, wave 0) from a measurement
build of the library (csrc/upconv_fused_h16.hip compiled with -DFV_PROF=1, passed as WDG_LIB):
    WDG_LIB=.../libwdgan_fvP.so python tools/prof_fused.py [bf16|fp16]"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402

NAMES = ["tile prologue (window load, staging, coefficients)", "weights of row 0: wait + staging", "barrier",
         "fragments to registers + MFMA phase of row 0", "rows 1 staged, MFMA row 1 + horizontal gather row 0",
         "barrier Z (H complete, next z complete)", "vertical gather + weights + barrier Y + MFMA / horizontal gather",
         "epilogue"]


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    from downscaling.engine import runtime
    from downscaling.gan.models import make_generator
    ops = runtime.get_ops()
    g = make_generator(96, 3, 20, 2, 24)
    g.graph_inference = False
    tiles = torch.randn(16, 24, 96, 96, 3, device=ops.device)
    noise = torch.randn(16, 24, 96, 96, 20, device=ops.device) * 0.1
    for _ in range(2):
        g([tiles, noise], precision=prec)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    ops.lib.wdg_fv_prof.restype = C.c_int
    assert ops.lib.wdg_fv_prof(out) == 0
    tot = sum(out)
    print(f"workgroup 0: {tot} clocks over its tiles")
    for n, v in zip(NAMES, out):
        print(f"  {n:40s} {v:10d}  {v / tot:6.1%}")


if __name__ == "__main__":
    main()
