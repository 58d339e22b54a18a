#!/usr/bin/env python3
"""The non-headline BASELINE.json configurations, run in fp32 on one MI355X (GPU box):
  configs[0]  generator-only forward, one 2-ch 16x16 ERA5 patch (x8 nearest) + 128x128 DEM -> 128x128 wind,
              checked against the CPU oracle (plumbing);
  configs[3]  tiled inference of a synthetic 1200x1200x24h field with the shipped network shape
              G(96,3,20,2,T=24): 225 tiles, groups of 16, overlap blend — fp32 and bf16-operand paths;
  configs[4]  stochastic ensemble: 8 tiles x 64 noise realisations — fp32, fp16-operand (the configuration's own
              precision) and bf16-operand paths.
Prints one JSON object."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from downscaling.engine import runtime
    import downscaling.api as api
    from downscaling.gan.models import make_generator
    ops = runtime.get_ops()
    out = {}
    # ---- configs[0]
    rng = np.random.default_rng(0)
    wind = np.repeat(np.repeat(rng.standard_normal((1, 1, 16, 16, 2)), 8, axis=2), 8, axis=3)
    dem = np.random.default_rng(1).standard_normal((1, 1, 128, 128, 1))
    noise = 0.1 * np.random.default_rng(2).standard_normal((1, 1, 128, 128, 20))
    g = make_generator(128, 3, 20, 2, 1)
    image = np.concatenate([wind, dem], -1)
    for _ in range(3):   # plan creation, lazy kernel loading, allocator warm-up
        y = g([image, noise], training=False)
    ts = []
    for _ in range(20):     # per-call latency (host numpy in -> device tensor out), median of 20
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = g([image, noise], training=False)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    # (timing only: the parity of this configuration against the fp64 restatement is asserted by tests/test_configs_gpu.py —
    # measurement tools do not import oracle/)
    out["config0_generator_forward_128"] = {"ms": 1e3 * dt, "finite": bool(torch.isfinite(y).all())}
    del g
    # ---- configs[3]: 24 x 1200 x 1200 field
    network = api.get_network(allow_random_init=True, random_seed=5)
    fields = np.random.default_rng(3).standard_normal((24, 1200, 1200, 3)).astype(np.float32)
    fields[..., 2] = fields[..., 2] * 500 + 1200
    t0 = time.perf_counter()
    pred, cnt = api.predict_array(fields, overlap_factor=0.05, network=network, return_count=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ntiles = 15 * 15
    out["config3_tiled_inference_1200x1200x24_fp32"] = {"tiles": ntiles, "seconds_end_to_end": dt,
                                                        "tile_timesteps_per_s": ntiles * 24 / dt,
                                                        "covered_fraction": float((cnt[0] > 0).mean()),
                                                        "finite": bool(np.isfinite(pred[:, cnt[0] > 0]).all())}
    # generator-only part (tiles resident on the GPU)
    gen = network.generator
    tiles = torch.randn(16, 24, 96, 96, 3, device=ops.device)
    nz = network.noise_generator(bs=16, channels=20)
    gen([tiles, nz])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        gen([tiles, network.noise_generator.lazy(bs=16, channels=20)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    gf = 3.799e9  # SURVEY §8 d: S=96, T=24 algorithmic forward FLOPs per tile-timestep
    out["config3_generator_only_16tiles_T24"] = {"ms_per_group": 1e3 * dt, "tile_timesteps_per_s": 16 * 24 / dt,
                                                 "tflops": 16 * 24 * gf / dt * 1e-12}
    gen.inference_precision = "bf16"
    ref32 = gen([tiles, nz], precision="fp32")
    got16 = gen([tiles, nz])
    rel = float((got16 - ref32).abs().max() / ref32.abs().max())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        gen([tiles, network.noise_generator.lazy(bs=16, channels=20)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    out["config3_generator_only_16tiles_T24_bf16"] = {"ms_per_group": 1e3 * dt, "tile_timesteps_per_s": 16 * 24 / dt,
                                                      "rel_err_vs_fp32_path": rel}
    runs = []
    for _ in range(2):                 # single-shot wall times of a 0.1 s job scatter (allocator, first use of a graph): both are kept
        t0 = time.perf_counter()
        pred16 = api.predict_array(fields, overlap_factor=0.05, network=network)
        torch.cuda.synchronize()
        runs.append(time.perf_counter() - t0)
    out["config3_tiled_inference_1200x1200x24_bf16"] = {"seconds_end_to_end": min(runs), "seconds_each_run": [round(r, 4) for r in runs],
                                                        "finite": bool(np.isfinite(pred16[:, cnt[0] > 0]).all())}
    phases = {}
    api.predict_array(fields, overlap_factor=0.05, network=network, timings=phases)    # (synchronises after every phase)
    out["config3_tiled_inference_1200x1200x24_bf16"]["phase_seconds"] = {k: round(v, 4) for k, v in phases.items() if k != 'laps'}
    out["config3_tiled_inference_1200x1200x24_bf16"]["laps"] = phases['laps']
    gen.inference_precision = "fp32"
    # ---- configs[4]: 8 tiles x 64 noise realisations
    tiles8 = tiles[:8]
    t0 = time.perf_counter()
    ens = []
    for r in range(64):
        ens.append(gen([tiles8, network.noise_generator.lazy(bs=8, channels=20)]))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    spread = float(torch.stack(ens).std(0).mean())
    out["config4_ensemble_8tiles_x64_fp32_1gpu"] = {"seconds": dt, "realisations_per_s": 64 * 8 / dt, "mean_spread": spread}
    ref_ens = torch.stack(ens).mean(0)
    for prec in ("fp16", "bf16"):        # configs[4] names the fp16 MFMA path; bf16 beside it
        gen.inference_precision = prec
        t0 = time.perf_counter()
        ens16 = []
        for r in range(64):
            ens16.append(gen([tiles8, network.noise_generator.lazy(bs=8, channels=20)]))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        dev = float((torch.stack(ens16).mean(0) - ref_ens).abs().max() / ref_ens.abs().max())
        out[f"config4_ensemble_8tiles_x64_{prec}_1gpu"] = {"seconds": dt, "realisations_per_s": 64 * 8 / dt,
                                                             "ensemble_mean_rel_dev_vs_fp32_ensemble": dev}
    gen.inference_precision = "fp32"
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:      # also to a file: stdout carries the reference-style progress prints of predict()
        Path(sys.argv[1]).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
