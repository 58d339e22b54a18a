"""The tiled inference driver (`predict_array` = array core of api.predict, /root/reference/src/downscaling/
api.py:96-151) against a literal restatement of the reference's steps — elevation/1e3, tile plan, per-tile
latitude flip with the sy == 0 off-by-one, nanmean/nanstd over axes (0,1,2), groups of 16 with fresh noise,
2-px crop, pandas concat + groupby(time, lat, lon).mean() — using the oracle generator on the same weights
and the same Philox noise.  CPU: patched small constants on the oracle backend; GPU: shipped constants
(G(96,3,20,2,T=24)) on the HIP backend."""
import math

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import torch_model as TM
from oracle.torch_backend import TorchOps, philox_normal_np


def reference_predict(fields, weights, seed, api, overlap_factor):
    """api.py:96-151 with numpy/pandas (xarray only carried coordinates there)."""
    IMG, SEQ, NZ, STD = api.IMG_SIZE, api.SEQUENCE_LENGTH, api.NOISE_CHANNELS, api.NOISE_STD
    ds = np.asarray(fields, dtype=np.float32).copy()
    ds[..., 2] = ds[..., 2] / 1e3
    T, Hh, Ww = ds.shape[:3]
    plan = api.tile_plan(Hh, Ww, T, overlap_factor)
    keys, tiles = [], []
    for sx in plan["slices_start_x"]:
        for sy in plan["slices_start_y"]:
            for k in range(plan["ntimeseq"]):
                lat = slice(sy + IMG - 1, sy - 1, -1) if sy != 0 else slice(IMG, 0, -1)
                tiles.append(ds[k * SEQ:(k + 1) * SEQ, lat, sx:sx + IMG])
                keys.append((sx, sy, k))
    tensors = np.stack(tiles, 0)
    tensors = (tensors - np.nanmean(tensors, axis=(0, 1, 2), keepdims=True)) / np.nanstd(tensors, axis=(0, 1, 2), keepdims=True)
    preds, offset = [], 0
    group = api.BATCH_SIZE * 2
    ngroups = math.ceil(len(tiles) / group)
    for t in range(ngroups):
        x = tensors[t * group:(t + 1) * group]
        # the driver draws the group's noise straight into the generator's time-major input buffer (FlexibleNoiseGenerator.lazy):
        # stream order (time, tile, x, y, channel) at the batch size of the call — a short last group runs padded to a full one
        bc = group if (x.shape[0] < group and ngroups > 1) else x.shape[0]
        n = bc * SEQ * IMG * IMG * NZ
        noise = (philox_normal_np(n, seed, offset) * STD).reshape(SEQ, bc, IMG, IMG, NZ).transpose(1, 0, 2, 3, 4)[:x.shape[0]]
        noise = np.ascontiguousarray(noise)
        offset += (n + 3) // 4
        with torch.no_grad():
            preds.append(TM.generator_forward(weights, torch.tensor(x, dtype=torch.float64), torch.tensor(noise), False).numpy())
    predictions = np.concatenate(preds, 0)
    frames = []
    for i, (sx, sy, k) in enumerate(keys):
        lat_idx = (np.arange(sy + IMG - 1, sy - 1, -1) if sy != 0 else np.arange(IMG, 0, -1))[2:-2]
        lon_idx = np.arange(sx, sx + IMG)[2:-2]
        tt, la, lo = np.meshgrid(np.arange(k * SEQ, (k + 1) * SEQ), lat_idx, lon_idx, indexing="ij")
        p = predictions[i][:, 2:-2, 2:-2]
        frames.append(pd.DataFrame({"time": tt.ravel(), "lat": la.ravel(), "lon": lo.ravel(),
                                    "u10": p[..., 0].ravel(), "v10": p[..., 1].ravel()}))
    return pd.concat(frames).groupby(["time", "lat", "lon"]).mean()


def _check(api, network, fields, overlap_factor, tol):
    weights = {k: torch.tensor(v, dtype=torch.float64) for k, v in network.generator.get_weights_dict().items()}
    seed = network.noise_generator.prng.seed
    assert network.noise_generator.prng.offset == 0
    out, cnt = api.predict_array(fields, overlap_factor=overlap_factor, network=network, return_count=True)
    ref = reference_predict(fields, weights, seed, api, overlap_factor)
    idx = ref.index.to_frame().to_numpy()
    got = out[idx[:, 0], idx[:, 1], idx[:, 2]]
    want = ref[["u10", "v10"]].to_numpy()
    assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max())
    assert int((cnt > 0).sum()) == len(ref)                   # same set of covered pixels
    assert np.isnan(out[cnt == 0]).all()


def test_predict_array_cpu(monkeypatch):
    from downscaling.engine import runtime
    import downscaling.api as api
    runtime.set_ops(TorchOps(torch.float64))
    try:
        monkeypatch.setattr(api, "IMG_SIZE", 20)
        monkeypatch.setattr(api, "SEQUENCE_LENGTH", 2)
        monkeypatch.setattr(api, "NOISE_CHANNELS", 5)
        monkeypatch.setattr(api, "BATCH_SIZE", 2)
        network = api.get_network(allow_random_init=True, random_seed=11)
        rng = np.random.default_rng(0)
        fields = rng.standard_normal((5, 40, 50, 3)).astype(np.float32)
        fields[..., 2] = fields[..., 2] * 800 + 1500
        _check(api, network, fields, 0.3, 1e-6)   # outputs are stored as float32
    finally:
        runtime.set_ops(None)


@pytest.mark.gpu
def test_predict_array_gpu(hip_ops):
    from downscaling.engine import runtime
    import downscaling.api as api
    runtime.set_ops(hip_ops)
    network = api.get_network(allow_random_init=True, random_seed=12)
    assert (network.generator.net.S, network.generator.net.T) == (96, 24)
    rng = np.random.default_rng(1)
    fields = rng.standard_normal((24, 100, 110, 3)).astype(np.float32)
    fields[..., 2] = fields[..., 2] * 800 + 1500
    _check(api, network, fields, 0.05, 1e-4)
