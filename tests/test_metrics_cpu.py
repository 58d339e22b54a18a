"""Evaluation metrics (gan/metrics.py:32-187 of the reference) against numpy restatements of their formulas."""
import numpy as np
import torch

from downscaling.gan import metrics as M


def _data(seed=0, B=3, T=2, S=20):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((B, T, S, S, 2)) * 4, rng.standard_normal((B, T, S, S, 2)) * 4


def test_extreme_ws_acd_lsd_match_numpy():
    a, b = _data()
    ta, tb = torch.tensor(a), torch.tensor(b)
    sq = a ** 2
    w = sq / sq.sum()
    np.testing.assert_allclose(M.extreme_weighted_rmse(ta, tb).numpy(), np.sqrt((w * (a - b) ** 2).sum(axis=(1, 2, 3, 4))), rtol=1e-12)
    ws = lambda x: np.sqrt(x[..., 0] ** 2 + x[..., 1] ** 2)
    np.testing.assert_allclose(M.wind_speed_rmse(ta, tb).numpy(), np.sqrt(((ws(a) - ws(b)) ** 2).mean(axis=(1, 2, 3))), rtol=1e-12)
    cos = (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))
    np.testing.assert_allclose(M.angular_cosine_distance(ta, tb).numpy(), (np.arccos(np.clip(cos, -1, 1)) / np.pi).mean(axis=(1, 2, 3)), rtol=1e-10)
    np.testing.assert_allclose(M.opposite_cosine_similarity(ta, tb).numpy(), (0.5 * (1 - cos)).mean(axis=(1, 2, 3)), rtol=1e-10)
    pw = lambda x: np.abs(np.fft.rfft2(np.transpose(x, (0, 1, 4, 2, 3)))) ** 2
    ratio = np.transpose((pw(a) + 1e-7) / (pw(b) + 1e-7), (0, 1, 3, 4, 2))
    np.testing.assert_allclose(M.log_spectral_distance(ta, tb).numpy(), np.sqrt(((10 * np.log10(ratio)) ** 2).mean(axis=(1, 2, 3, 4))), rtol=1e-9)
    # identical fields: zero distance everywhere
    for fn in (M.extreme_weighted_rmse, M.wind_speed_rmse, M.log_spectral_distance, M.wind_speed_weighted_rmse):
        assert float(fn(ta, ta).abs().max()) < 1e-12


def test_spatial_ks_and_wrappers():
    a, b = _data(1, B=2, T=1, S=20)
    ta, tb = torch.tensor(a), torch.tensor(b)
    img = M.spatially_convolved_ks_stat(ta, tb)          # patch = 20 // 10 = 2 -> 19 x 19 positions
    assert tuple(img.shape) == (19, 19)
    pts = np.linspace(-30, 30, 100)
    acc = np.zeros((2, 2, 19, 19))
    for ch in range(2):
        for n in range(2):
            for i in range(19):
                for j in range(19):
                    p1, p2 = a[n, 0, i:i + 2, j:j + 2, ch].ravel(), b[n, 0, i:i + 2, j:j + 2, ch].ravel()
                    acc[ch, n, i, j] = max(abs((p1 <= p).mean() - (p2 <= p).mean()) for p in pts)
    np.testing.assert_allclose(img.numpy(), acc.mean(axis=(0, 1)), atol=1e-12)
    assert float(M.spatially_convolved_ks_stat(ta, ta).abs().max()) == 0.0
    names = [m().name for m in (M.AngularCosineDistance, M.LogSpectralDistance, M.WeightedRMSEForExtremes,
                                M.WindSpeedWeightedRMSE, M.SpatialKS, M.WindSpeedRMSE)]
    assert names == ["acd", "lsd", "extreme_rmse", "ws_weighted_rmse", "spatial_ks", "ws_rmse"]
    m = M.WindSpeedRMSE()
    m.update_state(ta, tb)
    m.update_state(ta, ta)
    assert abs(m.result() - float(M.wind_speed_rmse(ta, tb).sum()) / 4) < 1e-12
