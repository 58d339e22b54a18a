"""Metrics of the reference (/root/reference/src/downscaling/gan/metrics.py:8-187): running means of the
discriminator scores, the wind-speed-weighted RMSE that can serve as the generator's content
(reconstruction-slot) loss, and the evaluation metrics `get_network` wires in (angular cosine distance,
log-spectral distance, extreme-weighted RMSE, wind-speed RMSE, spatial KS) — computed by the fused HIP reductions
of csrc/metrics.hip through the operator backend (no torch arithmetic on the product path, except the
differentiable form the reconstruction-loss slot needs)."""
import math

import torch


class Mean:
    def __init__(self, name="mean", **kwargs):
        self.name = name
        self.reset_states()

    def reset_states(self):
        self.total, self.count = 0.0, 0.0

    def update_state(self, values, sample_weight=None):
        values = torch.as_tensor(values).double()
        if sample_weight is not None:
            sw = torch.as_tensor(sample_weight, dtype=values.dtype, device=values.device).reshape(-1, *[1] * (values.dim() - 1))
            self.total += float((values * sw).sum())
            self.count += float(sw.expand_as(values).sum())
        else:
            self.total += float(values.sum())
            self.count += float(values.numel())

    def result(self):
        return self.total / self.count if self.count else 0.0


class discriminator_score_real(Mean):
    def __init__(self, name='d_real', **kwargs):
        super().__init__(name=name, **kwargs)

    def update_state(self, real_output, fake_output, sample_weight=None):
        return super().update_state(real_output, sample_weight)


class discriminator_score_fake(Mean):
    def __init__(self, name='d_fake', **kwargs):
        super().__init__(name=name, **kwargs)

    def update_state(self, real_output, fake_output, sample_weight=None):
        return super().update_state(fake_output, sample_weight)


# ---- the metric functions of gan/metrics.py:32-187 on the operator backend -------------------------------------------
# Every function takes (real_output, fake_output) of shape (B, T, H, W, 2) and returns the reference's per-sample vector
# (or the patch-position image for the spatial KS).  The arithmetic is the HIP kernels of csrc/metrics.hip: one fused
# pass yields the sums of all pointwise metrics, so a train step with the five compiled metrics reads the two fields
# once for them; the pass result is shared by the sibling metrics of one `_Metrics.update_state` call (pointwise_scope).
import numpy as np

from downscaling.engine import runtime

KS_POINTS = np.linspace(-30., 30., 100)          # metrics.py:156
_cache = {}


def _dev_pair(real_output, fake_output):
    from downscaling.gan.models import _to_dev
    ops = runtime.get_ops()
    return ops, _to_dev(real_output, ops), _to_dev(fake_output, ops)


class pointwise_scope:
    """While active (one `_Metrics.update_state` call), the fused pointwise pass is evaluated once per (real, fake)
    pair and shared by the sibling metrics.  The cache holds the two tensor OBJECTS and is matched with `is`, and it is
    dropped when the scope closes: nothing is ever keyed on id() / data_ptr() / version, which the caching allocator
    and CPython both recycle from one train step to the next (raw HIP kernels do not bump torch's version counter)."""

    def __enter__(self):
        _cache["depth"] = _cache.get("depth", 0) + 1
        return self

    def __exit__(self, *exc):
        _cache["depth"] -= 1
        if _cache["depth"] == 0:
            _cache.pop("pointwise", None)
        return False


def _pointwise(real_output, fake_output):
    """(sums [B, 6], elements per sample) of the fused pointwise pass; shared inside a `pointwise_scope` only."""
    scoped = _cache.get("depth", 0) > 0
    if scoped:
        hit = _cache.get("pointwise")
        if hit is not None and hit[0] is real_output and hit[1] is fake_output:
            return hit[2]
    ops, r, f = _dev_pair(real_output, fake_output)
    if r.shape[-1] != 2:
        raise ValueError("the wind metrics need the two wind components on the last axis")
    res = (ops.metrics_pointwise(r, f), r[0].numel() // 2, r.dtype)
    if scoped:
        _cache["pointwise"] = (real_output, fake_output, res)
    return res


def _ws_weighted_rmse_autograd(real_output, fake_output):
    """Differentiable form for the reconstruction-loss slot (ganbase.py:57-61 takes its gradient w.r.t. the generated
    winds): the same formula in torch tensor ops."""
    u, v = real_output[..., 0], real_output[..., 1]
    u_hat, v_hat = fake_output[..., 0], fake_output[..., 1]
    est, rea = torch.sqrt(u_hat ** 2 + v_hat ** 2), torch.sqrt(u ** 2 + v ** 2)
    beta = (4 + rea) / (4 + est)
    tau = torch.where(est >= rea, torch.full_like(u, 0.425), torch.full_like(u, 1 - 0.425))
    result = tau * ((u_hat - beta * u) ** 2 + (v_hat - beta * v) ** 2)
    result = torch.where(torch.isnan(result), torch.zeros_like(result), result)
    return torch.sqrt(torch.mean(result, dim=(1, 2, 3)))


def wind_speed_weighted_rmse(real_output, fake_output):
    """metrics.py:32-45 (epsilon = 4, t = 0.425: J. Dujardin's thesis)."""
    if torch.is_tensor(fake_output) and fake_output.requires_grad:
        return _ws_weighted_rmse_autograd(torch.as_tensor(real_output, dtype=fake_output.dtype, device=fake_output.device), fake_output)
    sums, n, dt = _pointwise(real_output, fake_output)
    return torch.sqrt(sums[:, 0] / n).to(dt)


def wind_speed_rmse(real_output, fake_output):
    """metrics.py:79-88."""
    sums, n, dt = _pointwise(real_output, fake_output)
    return torch.sqrt(sums[:, 1] / n).to(dt)


def angular_cosine_distance(real_output, fake_output):
    """metrics.py:94-101."""
    sums, n, dt = _pointwise(real_output, fake_output)
    return (sums[:, 2] / n).to(dt)


def opposite_cosine_similarity(real_output, fake_output):
    """metrics.py:103-105."""
    sums, n, dt = _pointwise(real_output, fake_output)
    return (sums[:, 3] / n).to(dt)


def extreme_weighted_rmse(real_output, fake_output):
    """metrics.py:66-73: squared errors weighted by real^2 / sum(real^2) over the WHOLE batch (divide_no_nan)."""
    sums, n, dt = _pointwise(real_output, fake_output)
    total = sums[:, 4].sum()
    ok = torch.isfinite(total) & (total != 0)
    return torch.where(ok, torch.sqrt(sums[:, 5] / torch.where(ok, total, torch.ones_like(total))), torch.zeros_like(sums[:, 5])).to(dt)


def log_spectral_distance(real_output, fake_output):
    """metrics.py:121-137.  The reference applies tf.signal.rfft2d to the (B, T, H, W, C) tensor itself, i.e. over its
    last two axes (W, C); the transposes it wraps around the absolute value cancel.  Reproduced as written."""
    ops, r, f = _dev_pair(real_output, fake_output)
    sums, n = ops.lsd_sums(r, f, 1e-7)                       # tf.keras.backend.epsilon()
    lsd = torch.sqrt(sums / n)
    return torch.where(torch.isnan(lsd), torch.zeros_like(lsd), lsd).to(r.dtype)


def spatially_convolved_ks_stat(real_output, fake_output, patch_size=None):
    """metrics.py:164-187 (patch = image size // 10, stride 1): image of the mean KS statistic per patch position."""
    ops, r, f = _dev_pair(real_output, fake_output)
    patch_size = patch_size or f.shape[2] // 10
    return ops.spatial_ks(r, f, int(patch_size), KS_POINTS).to(r.dtype)


def rmse_from_xarray(real_output, fake_output):
    """metrics.py:181-187 (numpy, evaluation scripts)."""
    real_output, fake_output = np.asarray(real_output), np.asarray(fake_output)
    u, v = real_output[..., 0], real_output[..., 1]
    u_hat, v_hat = fake_output[..., 0], fake_output[..., 1]
    return np.sqrt(np.mean((u - u_hat) ** 2 + (v - v_hat) ** 2, axis=(1, 2, 3)))


class WindSpeedWeightedRMSE(Mean):
    def __init__(self, name='ws_weighted_rmse', **kwargs):
        super().__init__(name=name, **kwargs)

    def update_state(self, y_true, y_pred, sample_weight=None):
        return super().update_state(wind_speed_weighted_rmse(y_true, y_pred), sample_weight)


class MeanMetricWrapper(Mean):
    """tfa.metrics.MeanMetricWrapper: running mean of fn(y_true, y_pred)."""

    def __init__(self, fn, name=None, **kwargs):
        super().__init__(name=name or fn.__name__)
        self._fn = fn

    def update_state(self, y_true, y_pred, sample_weight=None):
        return super().update_state(self._fn(y_true, y_pred), sample_weight if sample_weight is None or
                                    self._fn is not spatially_convolved_ks_stat else None)


WeightedRMSEForExtremes = lambda: MeanMetricWrapper(extreme_weighted_rmse, name='extreme_rmse')  # noqa: E731
WindSpeedRMSE = lambda: MeanMetricWrapper(wind_speed_rmse, name='ws_rmse')  # noqa: E731
AngularCosineDistance = lambda: MeanMetricWrapper(angular_cosine_distance, name='acd')  # noqa: E731
LogSpectralDistance = lambda: MeanMetricWrapper(log_spectral_distance, name='lsd')  # noqa: E731
SpatialKS = lambda: MeanMetricWrapper(spatially_convolved_ks_stat, name='spatial_ks')  # noqa: E731
