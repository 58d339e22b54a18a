#!/usr/bin/env python3
"""Generator-only inference forward of the shipped shape G(96,3,20,2,T=24) on 16 tiles (one predict() group), eager
launches, for `rocprofv3 --kernel-trace --stats`:  python tools/prof_infer.py [fp32|bf16|fp16] [reps] [key=int ...]
(trailing key=int pairs go to wdg_set_tuning before anything is planned: A/B runs)"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import time  # noqa: E402

import torch  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    from downscaling.engine import runtime
    from downscaling.gan.models import make_generator
    ops = runtime.get_ops()
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        assert ops.lib.wdg_set_tuning(k.encode(), int(v)) == 0, kv
    g = make_generator(96, 3, 20, 2, 24)
    g.graph_inference = False
    tiles = torch.randn(16, 24, 96, 96, 3, device=ops.device)
    noise = torch.randn(16, 24, 96, 96, 20, device=ops.device) * 0.1
    for _ in range(2):
        g([tiles, noise], precision=prec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g([tiles, noise], precision=prec)
    torch.cuda.synchronize()
    print(f"{prec} {sys.argv[3:]}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per 16-tile group (eager launches)")


if __name__ == "__main__":
    main()
