#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU: api.predict_ensemble(8 tiles, 64 realisations), fp16 operands.
    WDG_ENSEMBLE_TILES=<tiles per forward> python tools/perf_ensemble.py [fp16|bf16|fp32]"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402

import downscaling.api as api  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    network = api.get_network(allow_random_init=True, random_seed=5)
    dev = network.generator.ops.device
    tiles = torch.randn(8, api.SEQUENCE_LENGTH, api.IMG_SIZE, api.IMG_SIZE, 3, device=dev)
    api.predict_ensemble(tiles, 16, network=network, precision=prec)      # warm-up: plans, graphs
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ens = api.predict_ensemble(tiles, 64, network=network, precision=prec)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[1]
    print(f"{prec} WDG_ENSEMBLE_TILES={os.environ.get('WDG_ENSEMBLE_TILES', 'default')}: {64 * 8 / dt:.0f} realisations/s "
          f"({1e3 * dt:.1f} ms for 64 x 8), finite={bool(torch.isfinite(ens).all())}")


if __name__ == "__main__":
    main()
