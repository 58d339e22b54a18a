"""Metrics of the reference (/root/reference/src/downscaling/gan/metrics.py:8-187): running means of the
discriminator scores, the wind-speed-weighted RMSE that can serve as the generator's content
(reconstruction-slot) loss, and the evaluation metrics `get_network` wires in (angular cosine distance,
log-spectral distance, extreme-weighted RMSE, wind-speed RMSE, spatial KS)."""
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


def wind_speed_weighted_rmse(real_output, fake_output):
    # Only for cases where we output both wind speed components
    u, v = real_output[..., 0], real_output[..., 1]
    u_hat, v_hat = fake_output[..., 0], fake_output[..., 1]
    estimated_wind_speed = torch.sqrt(u_hat ** 2 + v_hat ** 2)
    realized_wind_speed = torch.sqrt(u ** 2 + v ** 2)
    epsilon = 4  # See Jerome Dujardin thesis
    t = 0.425  # See Jerome Dujardin thesis
    beta = (epsilon + realized_wind_speed) / (epsilon + estimated_wind_speed)
    tau = torch.where(estimated_wind_speed >= realized_wind_speed, torch.full_like(u, t), torch.full_like(u, 1 - t))
    result = tau * ((u_hat - beta * u) ** 2 + (v_hat - beta * v) ** 2)
    result = torch.where(torch.isnan(result), torch.zeros_like(result), result)
    return torch.sqrt(torch.mean(result, dim=(1, 2, 3)))


class WindSpeedWeightedRMSE(Mean):
    def __init__(self, name='ws_weighted_rmse', **kwargs):
        super().__init__(name=name, **kwargs)

    def update_state(self, y_true, y_pred, sample_weight=None):
        return super().update_state(wind_speed_weighted_rmse(y_true, y_pred), sample_weight)


# ---- the remaining evaluation metrics of the reference (gan/metrics.py:66-187), torch tensor ops ----------
# They are outside the timed GAN update: evaluation of (real, generated) winds, wired into
# `get_network` (api.py:77-81).  Plain torch reductions / FFT on whatever device the tensors live on.
def _l2_normalize(x, dim=-1, eps=1e-12):
    return x * torch.rsqrt(torch.clamp((x * x).sum(dim, keepdim=True), min=eps))


def cosine_similarity(y_true, y_pred, axis=-1):
    """tf.keras.losses.cosine_similarity: the NEGATIVE cosine similarity."""
    return -(_l2_normalize(y_true, axis) * _l2_normalize(y_pred, axis)).sum(axis)


def _divide_no_nan(a, b):
    return torch.where(b == 0, torch.zeros_like(a), a / torch.where(b == 0, torch.ones_like(b), b))


def extreme_weighted_rmse(real_output, fake_output):
    sq = real_output ** 2
    # Weights proportional to extremeness of winds
    weights = _divide_no_nan(sq, sq.sum())
    result = weights * (real_output - fake_output) ** 2
    result = torch.where(torch.isnan(result), torch.zeros_like(result), result)
    return torch.sqrt(result.sum(dim=(1, 2, 3, 4)))


def wind_speed_rmse(real_output, fake_output):
    # Only for cases where we output both wind speed components
    u, v = real_output[..., 0], real_output[..., 1]
    u_hat, v_hat = fake_output[..., 0], fake_output[..., 1]
    estimated_wind_speed = torch.sqrt(u_hat ** 2 + v_hat ** 2)
    realized_wind_speed = torch.sqrt(u ** 2 + v ** 2)
    result = (realized_wind_speed - estimated_wind_speed) ** 2
    result = torch.where(torch.isnan(result), torch.zeros_like(result), result)
    return torch.sqrt(result.mean(dim=(1, 2, 3)))


def angular_cosine_distance(real_output, fake_output):
    cos_sim = -cosine_similarity(real_output, fake_output)
    bounded_cos_sim = torch.clamp(cos_sim, -1, 1)
    acd = torch.acos(bounded_cos_sim) / math.pi
    return acd.mean(dim=(1, 2, 3))


def opposite_cosine_similarity(real_output, fake_output):
    cos_sim = .5 * (1 + cosine_similarity(real_output, fake_output))
    return cos_sim.mean(dim=(1, 2, 3))


def log_spectral_distance(real_output, fake_output):
    epsilon = 1e-7  # tf.keras.backend.epsilon()

    def power(x):   # rfft2d over (H, W) of (B, T, H, W, C)
        return (torch.fft.rfft2(x.permute(0, 1, 4, 2, 3)).abs() ** 2).permute(0, 1, 3, 4, 2)
    ratio = _divide_no_nan(power(real_output) + epsilon, power(fake_output) + epsilon)
    result = (10 * torch.log10(ratio)) ** 2
    lsd = torch.sqrt(result.mean(dim=(1, 2, 3, 4)))
    return torch.where(torch.isnan(lsd), torch.zeros_like(lsd), lsd)


def ks_stat_on_patch(patch1, patch2):
    """patches [..., n_samples]; sup over 100 points in [-30, 30] of |ECDF1 - ECDF2| (tfp Empirical.cdf)."""
    points = torch.linspace(-30., 30., 100, dtype=patch1.dtype, device=patch1.device)
    ks = torch.zeros(patch1.shape[:-1], dtype=patch1.dtype, device=patch1.device)
    for p in points:
        c1 = (patch1 <= p).to(patch1.dtype).mean(-1)
        c2 = (patch2 <= p).to(patch1.dtype).mean(-1)
        ks = torch.maximum(ks, (c1 - c2).abs())
    return ks


def spatially_convolved_ks_stat(real_output, fake_output, patch_size=None):
    patch_size = patch_size or fake_output.shape[2] // 10
    stats = []
    for time in range(fake_output.shape[1]):
        for ch in range(fake_output.shape[-1]):
            p1 = real_output[:, time, ..., ch].unfold(1, patch_size, 1).unfold(2, patch_size, 1).flatten(-2)
            p2 = fake_output[:, time, ..., ch].unfold(1, patch_size, 1).unfold(2, patch_size, 1).flatten(-2)
            stats.append(ks_stat_on_patch(p1, p2))
    return torch.stack(stats).mean(dim=(0, 1))


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
