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


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_predict_array_groups_per_launch_gpu(hip_ops, monkeypatch, precision):
    """Two predict() groups of 16 tiles per forward pass (the default, WDG_PREDICT_GROUPS=2) against one group per pass: every
    group keeps its own noise draw at its own place in the generator's stream (LazyGroupNoise, wdg_input_assemble_slots) and
    tiles are independent in inference mode, so the blended fields are the same bits — including a launch whose second group
    is short (50 tiles = 3 groups + 2 tiles -> launches of (16, 16) and (16, 2 + 14 padding))."""
    from downscaling.engine import runtime
    import downscaling.api as api
    runtime.set_ops(hip_ops)
    rng = np.random.default_rng(2)
    fields = rng.standard_normal((48, 400, 400, 3)).astype(np.float32)
    fields[..., 2] = fields[..., 2] * 800 + 1500
    outs = []
    for per in ("1", "2", "3"):
        monkeypatch.setenv("WDG_PREDICT_GROUPS", per)
        network = api.get_network(allow_random_init=True, random_seed=21)
        network.generator.inference_precision = precision
        out, cnt = api.predict_array(fields, overlap_factor=0.05, network=network, return_count=True)
        assert network.noise_generator.prng.offset == 4 * (16 * 24 * 96 * 96 * 20 // 4)      # four draws of a full group, whatever the launches
        outs.append(out)
    assert int((cnt > 0).sum()) > 0 and np.isfinite(outs[0][cnt > 0]).all()
    assert np.array_equal(outs[0], outs[1], equal_nan=True) and np.array_equal(outs[0], outs[2], equal_nan=True)


@pytest.mark.gpu
def test_tiling_kernels_gpu():
    """csrc/tiling.hip against the literal expressions of the driver (api.py:117-129 tile gather + np.nanmean / np.nanstd
    normalisation; api.py:139-150 cropped sum / count), NaNs in the field, overlapping tiles inside one group."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from downscaling.engine.hipops import HipOps
    ops = HipOps("cuda:0")
    dev = ops.device
    rng = np.random.default_rng(5)
    Ttot, LAT, LON, C, T, S = 4, 70, 90, 3, 2, 24
    field = rng.standard_normal((Ttot, LAT, LON, C)).astype(np.float32) * np.array([3, 5, 0.7], np.float32) + np.array([1, -2, 4], np.float32)
    field[rng.random(field.shape) < 0.01] = np.nan
    keys = [(sx, row0, k) for sx in (0, 20, 60) for row0 in (S, S + 13, LAT - 1) for k in (0, 1)]      # row0 = highest row of a tile
    tiles = np.stack([field[k * T:(k + 1) * T, row0 - np.arange(S)][:, :, sx:sx + S] for (sx, row0, k) in keys], 0)
    want = (tiles - np.nanmean(tiles, axis=(0, 1, 2), keepdims=True)) / np.nanstd(tiles, axis=(0, 1, 2), keepdims=True)
    keys4 = torch.tensor([[sx, row0, k, 0] for (sx, row0, k) in keys], dtype=torch.int32, device=dev)
    got = ops.tiles_gather_normalise(torch.tensor(field, device=dev), keys4, T, S).cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) < 2e-5 * np.nanmax(np.abs(want))
    # blend: 5 "real" tiles of a group of 7, crop 2
    B, n_real, crop = 7, 5, 2
    pred = rng.standard_normal((B, T, S, S, 2)).astype(np.float32)
    acc = np.zeros((Ttot, LAT, LON, 2))
    cnt = np.zeros((Ttot, LAT, LON), np.int32)
    for j, (sx, row0, k) in enumerate(keys[:n_real]):
        rows = (row0 - np.arange(S))[crop:-crop]
        acc[k * T:(k + 1) * T, rows[:, None], np.arange(sx + crop, sx + S - crop)[None, :]] += pred[j][:, crop:-crop, crop:-crop]
        cnt[k * T:(k + 1) * T, rows[:, None], np.arange(sx + crop, sx + S - crop)[None, :]] += 1
    acc_g = torch.zeros(Ttot, LAT, LON, 2, dtype=torch.float64, device=dev)
    cnt_g = torch.zeros(Ttot, LAT, LON, dtype=torch.int32, device=dev)
    ops.tiles_blend(torch.tensor(pred, device=dev), keys4[:B].contiguous(), n_real, acc_g, cnt_g, crop)
    assert np.array_equal(cnt_g.cpu().numpy(), cnt)
    assert np.abs(acc_g.cpu().numpy() - acc).max() < 1e-12
