#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels on the layer shapes of the headline workload (batch 32, 256x256):
A/B of the igemm pipeline variants (wdg_set_tuning) in ONE process, interleaved rounds, HIP-event timing.
Usage (GPU box):  python tools/perf_conv.py [reps]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402

from downscaling.engine.hipops import ConvGeom, HipOps  # noqa: E402

B = 32
# name, (n,H,W,cin,ld_in), cout, k, s, p, which
SHAPES = [
    ("G0 8x8s2 23->128 fwd", (B, 256, 256, 23, 24), 128, 8, 2, 3, "fwd"),
    ("G2 4x4s2 128->128 fwd", (B, 128, 128, 128, 160), 128, 4, 2, 1, "fwd"),
    ("G4 lstm 3x3 128->512 fwd", (B, 64, 64, 128, 192), 512, 3, 1, 1, "fwd"),
    ("G5 3x3 128->64 fwd", (B, 64, 64, 128, 128), 64, 3, 1, 1, "fwd"),
    ("D 7x7s3 32->64 fwd", (B, 256, 256, 32, 32), 64, 7, 3, 1, "fwd"),
    ("D 7x7s3 64->128 fwd", (B, 84, 84, 64, 64), 128, 7, 3, 1, "fwd"),
    ("D 7x7s3 128->256 fwd", (B, 27, 27, 128, 128), 256, 7, 3, 1, "fwd"),
    ("D 7x7s3 256->512 fwd", (B, 8, 8, 256, 256), 512, 7, 3, 1, "fwd"),
    ("G4 lstm 3x3 128->512 dgrad", (B, 64, 64, 128, 192), 512, 3, 1, 1, "dgrad"),
    ("G2 4x4s2 dgrad", (B, 128, 128, 128, 160), 128, 4, 2, 1, "dgrad"),
    ("D 7x7s3 32->64 dgrad", (B, 256, 256, 32, 32), 64, 7, 3, 1, "dgrad"),
    ("G0 wgrad", (B, 256, 256, 23, 24), 128, 8, 2, 3, "wgrad"),
    ("G4 lstm wgrad", (B, 64, 64, 128, 192), 512, 3, 1, 1, "wgrad"),
    ("G9 convT5x5 wgrad (16,160)", (B, 256, 256, 16, 16), 160, 5, 1, 2, "wgrad"),
    ("G9 convT5x5 dx (16->160 fwd)", (B, 256, 256, 16, 16), 160, 5, 1, 2, "fwd"),
    ("D 7x7s3 32->64 wgrad", (B, 256, 256, 32, 32), 64, 7, 3, 1, "wgrad"),
    # HBM-bound full-resolution layers with few channels (halo-tile kernels)
    ("D lstm_a 2->8 fwd", (B, 256, 256, 2, 4), 8, 3, 1, 1, "fwd"),
    ("D conv_a 2->16 fwd", (B, 256, 256, 2, 4), 16, 3, 1, 1, "fwd"),
    ("D conv_b 16->16 fwd", (B, 256, 256, 16, 16), 16, 3, 1, 1, "fwd"),
    ("D lstm_b 5->64 fwd", (B, 256, 256, 5, 8), 64, 3, 1, 1, "fwd"),
    ("G11 16->2 fwd", (B, 256, 256, 16, 16), 2, 3, 1, 1, "fwd"),
    ("D lstm_a 2->8 dgrad", (B, 256, 256, 2, 4), 8, 3, 1, 1, "dgrad"),
    ("D conv_a 2->16 dgrad", (B, 256, 256, 2, 4), 16, 3, 1, 1, "dgrad"),
    ("D conv_b 16->16 dgrad", (B, 256, 256, 16, 16), 16, 3, 1, 1, "dgrad"),
    ("D lstm_b 5->64 dgrad", (B, 256, 256, 5, 8), 64, 3, 1, 1, "dgrad"),
    ("G11 16->2 dgrad", (B, 256, 256, 16, 16), 2, 3, 1, 1, "dgrad"),
    ("D conv_a 2->16 wgrad", (B, 256, 256, 2, 4), 16, 3, 1, 1, "wgrad"),
    ("D conv_b 16->16 wgrad", (B, 256, 256, 16, 16), 16, 3, 1, 1, "wgrad"),
    ("D lstm_b 5->64 wgrad", (B, 256, 256, 5, 8), 64, 3, 1, 1, "wgrad"),
    ("G9 fused up+convT5x5 fwd", None, 0, 0, 0, 0, "upconv"),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    ops = HipOps("cuda:0")
    lib = ops.lib
    cases = []
    for name, shp, cout, k, s, p, which in SHAPES:
        if which == "upconv":
            xl = torch.randn(B, 128, 128, 160, device=ops.device)
            w = (torch.randn(5, 5, 16, 160, device=ops.device) * 0.05).contiguous()
            pk = ops.pack_weights(w)
            yo = torch.empty(B, 256, 256, 16, device=ops.device)
            g = ConvGeom(5, 5, 1, 2)
            cases.append((name, lambda xl=xl, pk=pk, yo=yo, g=g: ops.upconv_fwd(xl, pk, None, yo, g, act=True),
                          2.0 * B * 65536 * 16 * 25 * 160, "wdg_conv_halo_kernel<1>+up", (xl.numel() + yo.numel()) * 4))
            continue
        n, H, W, cin, ld = shp
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        xb = torch.randn(n, H, W, ld, device=ops.device)
        x = xb[..., :(cin + 3) // 4 * 4]
        if cin % 4:
            x[..., cin:] = 0
        y = torch.randn(n, Ho, Wo, (cout + 3) // 4 * 4, device=ops.device)
        w = (torch.randn(k, k, cin, cout, device=ops.device) * 0.05).contiguous()
        pk = ops.pack_weights(w)
        dw = torch.zeros_like(w)
        g = ConvGeom(k, k, s, p)
        flops = 2.0 * n * Ho * Wo * cout * k * k * cin
        if which == "fwd":
            fn = lambda x=x, pk=pk, y=y, g=g: ops.conv_fwd(x, pk, None, y, g, act=True)
        elif which == "dgrad":
            fn = lambda x=x, pk=pk, y=y, g=g: ops.conv_dgrad(y, pk, x, g)
        else:
            fn = lambda x=x, pk=pk, y=y, g=g, dw=dw: ops.conv_wgrad(x, y, pk, dw, g, accumulate=False)
        label = ops.conv_kernel_label(which, x, y, pk, g)
        nbytes = 4.0 * (n * H * W * ((cin + 3) // 4 * 4) + n * Ho * Wo * ((cout + 3) // 4 * 4))
        cases.append((name, fn, flops, label, nbytes))
    pipes = (0, 1, 2)   # columns: 0 = defaults, 1 = 128x160 tile off, 2 = K-loop variant 0 (plain single-stage loop)
    knobs = {0: ((b"tile160", 1), (b"igemm_pipe", 3)), 1: ((b"tile160", 0), (b"igemm_pipe", 3)), 2: ((b"tile160", 1), (b"igemm_pipe", 0))}
    times = {(c[0], pp): [] for c in cases for pp in pipes}
    for r in range(reps + 1):
        for pp in pipes:
            for k_, v_ in knobs[pp]:
                lib.wdg_set_tuning(k_, v_)
            for name, fn, flops, label, nbytes in cases:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    times[(name, pp)].append(e0.elapsed_time(e1))
    print(f"{'layer':32s} {'kernel':28s} " + " ".join(f"{n}: ms (TF/s)  " for n in ("default ", "no160   ", "pipe0   ")))
    for name, fn, flops, label, nbytes in cases:
        row = f"{name:32s} {label:28s} "
        for pp in pipes:
            t = sorted(times[(name, pp)])[len(times[(name, pp)]) // 2]
            row += f"{t:7.3f} ({flops / t * 1e-9:6.1f})   "
        t0 = sorted(times[(name, 0)])[len(times[(name, 0)]) // 2]
        print(row + f" in+out {nbytes / t0 * 1e-9:6.2f} TB/s")


if __name__ == "__main__":
    main()
