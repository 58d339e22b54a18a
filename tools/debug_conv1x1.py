#!/usr/bin/env python3
"""A thin convolution on a large map (the halo-tile kernel's territory) against torch-fp64:  python tools/debug_conv1x1.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402
import torch.nn.functional as Fn  # noqa: E402
from downscaling.engine.hipops import ConvGeom, HipOps  # noqa: E402

ops = HipOps("cuda:0")
dev = ops.device
g = torch.Generator().manual_seed(1)
for (n, H, W, cin, cout, k, p, ldx) in [(64, 64, 64, 100, 20, 1, 0, 100), (4, 64, 64, 100, 20, 1, 0, 100), (64, 64, 64, 52, 20, 1, 0, 52), (64, 64, 64, 50, 20, 1, 0, 52),
                                        (64, 64, 64, 100, 16, 1, 0, 100), (64, 64, 64, 48, 20, 3, 1, 48), (64, 64, 64, 24, 8, 1, 0, 24), (16, 64, 64, 100, 20, 1, 0, 100), (32, 64, 64, 100, 20, 1, 0, 100)]:
    x = torch.randn(n, H, W, ldx, generator=g, dtype=torch.float64)
    w = torch.randn(k, k, cin, cout, generator=g, dtype=torch.float64) * 0.1
    ref = Fn.conv2d(x[..., :cin].permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), padding=p).permute(0, 2, 3, 1)
    xg = x.float().to(dev)
    y = ops.zeros(n, H, W, cout)
    pk = ops.pack_weights(w.float().to(dev).contiguous())
    ops.conv_fwd(xg[..., :cin] if ldx != cin else xg, pk, None, y, ConvGeom(k, k, 1, p))
    err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    label = ops.conv_kernel_label("fwd", xg[..., :cin] if ldx != cin else xg, y, pk, ConvGeom(k, k, 1, p))
    print(f"n={n} {H}x{W} {k}x{k} cin={cin} (ld {ldx}) cout={cout}: {label:28s} rel err {err:.3e}")
