#!/usr/bin/env python3
"""Split-factor sweep for the split-K (implicit GEMM) and split-pixel (weight gradient) launches on the
headline layer shapes: evidence for the split heuristic in conv_igemm.hip (pick_split_model).
Usage (GPU box):  python tools/sweep_split.py [reps]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402

from downscaling.engine.hipops import ConvGeom, HipOps  # noqa: E402

B = 32
CASES = [
    # name, (n,H,W,cin,ld), cout, k, s, p, which, candidate splits
    ("G0 wgrad 23->128", (B, 256, 256, 23, 24), 128, 8, 2, 3, "wgrad", (21, 32, 42, 64, 85, 128)),
    ("G2 wgrad 128->128", (B, 128, 128, 128, 160), 128, 4, 2, 1, "wgrad", (16, 24, 32, 48, 64, 96)),
    ("G4 wgrad 128->512", (B, 64, 64, 128, 192), 512, 3, 1, 1, "wgrad", (7, 11, 14, 22, 28, 43)),
    ("D wgrad 32->64", (B, 256, 256, 32, 32), 64, 7, 3, 1, "wgrad", (20, 40, 60, 80, 120, 160)),
    ("D wgrad 64->128", (B, 84, 84, 64, 64), 128, 7, 3, 1, "wgrad", (5, 10, 15, 20, 30, 45)),
    ("D wgrad 128->256", (B, 27, 27, 128, 128), 256, 7, 3, 1, "wgrad", (1, 2, 3, 4, 6, 8)),
    ("D wgrad 256->512", (B, 8, 8, 256, 256), 512, 7, 3, 1, "wgrad", (1, 2, 4)),
    ("D fwd 64->128", (B, 84, 84, 64, 64), 128, 7, 3, 1, "fwd", (1, 2, 3, 4, 5, 6, 8, 11)),
    ("D fwd 128->256", (B, 27, 27, 128, 128), 256, 7, 3, 1, "fwd", (4, 8, 12, 16, 24, 32)),
    ("D fwd 256->512", (B, 8, 8, 256, 256), 512, 7, 3, 1, "fwd", (32, 64, 98, 128, 196)),
    ("D dgrad 64->128", (B, 84, 84, 64, 64), 128, 7, 3, 1, "dgrad", (1, 2, 3, 4)),
    ("D dgrad 128->256", (B, 27, 27, 128, 128), 256, 7, 3, 1, "dgrad", (1, 2, 3, 4, 6)),
    ("D dgrad 256->512", (B, 8, 8, 256, 256), 512, 7, 3, 1, "dgrad", (2, 4, 8, 16, 32)),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    ops = HipOps("cuda:0")
    lib = ops.lib
    for name, shp, cout, k, s, p, which, cands in CASES:
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
        key = {"fwd": b"force_fwd_split", "dgrad": b"force_dgrad_split", "wgrad": b"force_wgrad_split"}[which]
        row = f"{name:20s}"
        for c in (0,) + tuple(cands):
            lib.wdg_set_tuning(key, c)
            ops._plans.clear()
            if which == "fwd":
                fn = lambda: ops.conv_fwd(x, pk, None, y, g, act=True)
            elif which == "dgrad":
                fn = lambda: ops.conv_dgrad(y, pk, x, g)
            else:
                fn = lambda: ops.conv_wgrad(x, y, pk, dw, g, accumulate=False)
            fn()
            info = ops._plan(x, y, cin, cout, g)[2]
            chosen = info[{"fwd": 2, "dgrad": 5, "wgrad": 7}[which]]
            ts = []
            for _ in range(reps):
                xb.normal_()   # flush-ish: touch another 100s of MB between runs
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            t = sorted(ts)[len(ts) // 2]
            row += f"  s={chosen:<3d}{'*' if c == 0 else ' '} {t * 1e3:6.0f}us {flops / t * 1e-9:5.1f}"
        lib.wdg_set_tuning(key, 0)
        print(row, flush=True)


if __name__ == "__main__":
    main()
