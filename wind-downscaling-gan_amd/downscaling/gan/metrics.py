"""The metrics that touch the train step (/root/reference/src/downscaling/gan/metrics.py:8-45):
running means of the discriminator scores and the wind-speed-weighted RMSE that can serve as the
generator's content (reconstruction-slot) loss.  The remaining evaluation metrics of the reference
(log-spectral distance, spatial KS, ...) are outside the hot path."""
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
