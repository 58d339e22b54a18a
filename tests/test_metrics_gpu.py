"""csrc/metrics.hip (the fused evaluation-metric reductions, reference gan/metrics.py:32-187) against the oracle
restatement (oracle/torch_backend.py, fp64) on the same inputs, through the package's metric functions; and
GAN.train_step with the five metrics of get_network compiled in (api.py:77-81, ganbase.py:71) on the device."""
import numpy as np
import pytest
import torch

from downscaling.engine import runtime
from downscaling.gan import metrics as M
from oracle.torch_backend import TorchOps

pytestmark = pytest.mark.gpu


def _fields(seed, B, T, S):
    rng = np.random.default_rng(seed)
    real = (rng.standard_normal((B, T, S, S, 2)) * 4).astype(np.float32)
    fake = (real + rng.standard_normal((B, T, S, S, 2)) * 2).astype(np.float32)
    return real, fake


def _both(fn, real, fake, hip_ops, **kw):
    runtime.set_ops(hip_ops)
    got = fn(torch.from_numpy(real), torch.from_numpy(fake), **kw)
    assert got.is_cuda
    runtime.set_ops(TorchOps(torch.float64))
    try:
        want = fn(torch.from_numpy(real).double(), torch.from_numpy(fake).double(), **kw)
    finally:
        runtime.set_ops(hip_ops)
    return got.double().cpu().numpy(), want.numpy()


@pytest.mark.parametrize("B,T,S", [(3, 2, 20), (2, 1, 96), (5, 3, 33)])
def test_pointwise_and_spectral_metrics(hip_ops, B, T, S):
    real, fake = _fields(B * 100 + S, B, T, S)
    for fn, tol in ((M.wind_speed_weighted_rmse, 2e-6), (M.wind_speed_rmse, 2e-6), (M.angular_cosine_distance, 1e-5),
                    (M.opposite_cosine_similarity, 2e-6), (M.extreme_weighted_rmse, 2e-6), (M.log_spectral_distance, 2e-5)):
        got, want = _both(fn, real, fake, hip_ops)
        assert got.shape == (B,)
        np.testing.assert_allclose(got, want, rtol=tol, atol=tol * 1e-2, err_msg=fn.__name__)
    # NaN handling: masked terms (tf.where(is_nan)), NaN-poisoned weight sum -> 0
    bad = real.copy()
    bad[0, 0, 1, 2, 0] = np.nan
    for fn in (M.wind_speed_weighted_rmse, M.wind_speed_rmse, M.extreme_weighted_rmse):
        got, want = _both(fn, bad, fake, hip_ops)
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-9, err_msg=fn.__name__)
    # zero vectors: keras' l2_normalize clamp keeps the cosine finite
    z = real.copy()
    z[:, :, :3] = 0.0
    got, want = _both(M.angular_cosine_distance, z, fake, hip_ops)
    np.testing.assert_allclose(got, want, rtol=1e-5)


@pytest.mark.parametrize("B,T,S,patch", [(2, 1, 20, None), (2, 2, 96, None), (1, 1, 40, 7), (3, 1, 17, 17)])
def test_spatial_ks(hip_ops, B, T, S, patch):
    real, fake = _fields(7 + S, B, T, S)
    real[0, 0, 2, 3, 1] = 31.0            # beyond the last evaluation point
    fake[0, 0, 5, 5, 0] = -30.0           # exactly the first one
    fake[-1, -1, 4, 1, 1] = np.nan
    got, want = _both(M.spatially_convolved_ks_stat, real, fake, hip_ops, patch_size=patch)
    p = patch or S // 10
    assert got.shape == (S - p + 1, S - p + 1)
    # the statistic is a ratio of small integers: exact up to the fp64 mean
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-7)      # integer counts, fp64 mean; returned in the input dtype (fp32)
    same, _ = _both(M.spatially_convolved_ks_stat, real, real.copy(), hip_ops, patch_size=patch)
    assert float(np.abs(same).max()) == 0.0


def test_train_step_with_compiled_metrics(hip_ops):
    """GAN.compile(generator_metrics=[...the five of api.py:77-81...]) -> train_step returns g_<name> for each; values
    equal the oracle metrics of (high_res, last generated batch)."""
    from downscaling.data.data_generator import FlexibleNoiseGenerator
    from downscaling.gan import train
    from downscaling.gan.ganbase import GAN
    from downscaling.gan.models import make_discriminator, make_generator
    runtime.set_ops(hip_ops)
    B, T, S = 2, 2, 32
    g, d = make_generator(S, 3, 4, 2, T, feature_channels=32), make_discriminator(S, S, 3, 2, T, feature_channels=8)
    gan = GAN(g, d, FlexibleNoiseGenerator((B, T, S, S, 4), std=0.1, random_seed=3), n_critic=1)
    mets = [M.AngularCosineDistance(), M.LogSpectralDistance(), M.WeightedRMSEForExtremes(), M.WindSpeedWeightedRMSE(), M.SpatialKS()]
    gan.compile(train.generator_optimizer(), train.discriminator_optimizer(), generator_metrics=mets,
                discriminator_loss=train.discriminator_loss,
                metrics=[M.discriminator_score_fake(), M.discriminator_score_real()])
    rng = np.random.default_rng(0)
    sums, per_step = {}, []
    for step in range(3):
        # a NEW batch every step, uploaded from numpy (lands in the allocator block the previous step's batch just freed) and a
        # fresh `fake` tensor filled by a raw HIP kernel (version counter stays 0): the situation in which a cache keyed on
        # id() / data_ptr() / _version would hand back the previous step's sums (gan/metrics.py)
        low = rng.standard_normal((B, T, S, S, 3)).astype(np.float32)
        high = (rng.standard_normal((B, T, S, S, 2)) * (3 + step)).astype(np.float32)
        logs = gan.train_step((low, high))
        assert {"g_acd", "g_lsd", "g_extreme_rmse", "g_ws_weighted_rmse", "g_spatial_ks", "d_fake", "d_real"} <= set(logs)
        fake = torch.empty(B, T, S, S, 2, device=hip_ops.device)
        g.net.from_time_major(gan.engine.last_fake_tm, fake)
        fake = fake.cpu().numpy()
        runtime.set_ops(TorchOps(torch.float64))
        try:
            th, tf_ = torch.from_numpy(high).double(), torch.from_numpy(fake).double()
            now = {"g_acd": M.angular_cosine_distance(th, tf_).mean(), "g_lsd": M.log_spectral_distance(th, tf_).mean(),
                   "g_extreme_rmse": M.extreme_weighted_rmse(th, tf_).mean(), "g_ws_weighted_rmse": M.wind_speed_weighted_rmse(th, tf_).mean(),
                   "g_spatial_ks": M.spatially_convolved_ks_stat(th, tf_).mean()}
        finally:
            runtime.set_ops(hip_ops)
        per_step.append({k: float(v) for k, v in now.items()})
        for k, v in now.items():              # Keras metric objects are running means over the steps since the last reset
            sums[k] = sums.get(k, 0.0) + float(v)
            want = sums[k] / (step + 1)
            assert abs(float(logs[k]) - want) <= 2e-5 * max(1.0, abs(want)), (step, k, float(logs[k]), want)
    for k in ("g_acd", "g_extreme_rmse", "g_ws_weighted_rmse"):      # the batches differ: so must the per-step values
        assert abs(per_step[0][k] - per_step[1][k]) > 1e-3 * abs(per_step[0][k]), k


def test_pointwise_pass_is_not_cached_across_calls(hip_ops):
    """Outside a `_Metrics.update_state` scope nothing is cached; inside one, the sums are matched by object identity of live
    tensors only.  Overwriting a tensor in place through a raw kernel (no version bump) must change the result."""
    from downscaling.gan.metrics import pointwise_scope
    runtime.set_ops(hip_ops)
    dev = hip_ops.device
    real = torch.randn(2, 1, 16, 16, 2, device=dev)
    fake = torch.randn(2, 1, 16, 16, 2, device=dev)
    a = M.wind_speed_rmse(real, fake).clone()
    fake2 = torch.randn(2, 1, 16, 16, 2, device=dev)
    hip_ops.copy_channels(fake2.view(2, 16, 16, 2), fake.view(2, 16, 16, 2))     # raw HIP kernel: fake._version unchanged
    b = M.wind_speed_rmse(real, fake).clone()
    assert not torch.equal(a, b)
    with pointwise_scope():
        c = M.wind_speed_rmse(real, fake).clone()
        d = M.angular_cosine_distance(real, fake)                                  # shares the pass
        assert torch.equal(b, c) and d.shape == (2,)
    from downscaling.gan import metrics as mm
    assert "pointwise" not in mm._cache
