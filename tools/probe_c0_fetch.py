#!/usr/bin/env python3
"""FETCH_SIZE probe of one layer (generator c0: 8x8 stride 2, 23 -> 128, batch 32, 256x256) with the implicit GEMM's
2-D row tiles off / on:  rocprofv3 --pmc FETCH_SIZE -- python3 tools/probe_c0_fetch.py   (dispatch order: 3 x off, 3 x on)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402
from downscaling.engine.hipops import ConvGeom, HipOps  # noqa: E402

ops = HipOps("cuda:0")
x = torch.randn(32, 256, 256, 24, device=ops.device)
x[..., 23] = 0
w = (torch.randn(8, 8, 23, 128, device=ops.device) * 0.05).contiguous()
pk = ops.pack_weights(w)
y = torch.empty(32, 128, 128, 128, device=ops.device)
g = ConvGeom(8, 8, 2, 3)
for t2 in (0, 1):
    ops.lib.wdg_set_tuning(b"tile2d", t2)
    for _ in range(3):
        ops.conv_fwd(x, pk, None, y, g, act=True)
    torch.cuda.synchronize()
print("done")
