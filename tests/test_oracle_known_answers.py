"""Pins the oracle.  The reference has no tests, no golden vectors and TensorFlow cannot run here
("parity unpinned"), so the CPU restatement is pinned by (i) hand-computable known answers on the numpy
restatement (oracle/np_ops.py: impulse inputs, identity kernels, hard_sigmoid knots, bilinear edge rows,
transposed-conv phase placement, norms on constant / ramp inputs, first Adam step), and (ii) agreement of
the torch restatements (oracle/torch_model.py, oracle/torch_backend.py) with it."""
import numpy as np
import pytest
import torch

from oracle import np_ops as N
from oracle import torch_model as TM
from oracle.torch_backend import TorchOps, philox4x32_10


# ---------------- (i) known answers ----------------------------------------------------------------
def test_conv2d_impulse_and_identity():
    x = np.zeros((1, 5, 5, 1)); x[0, 2, 2, 0] = 1.0
    w = np.arange(9, dtype=float).reshape(3, 3, 1, 1)
    y = N.conv2d(x, w, None, 1, 1)
    # cross-correlation: an impulse reproduces the kernel flipped around the centre
    np.testing.assert_array_equal(y[0, 1:4, 1:4, 0], w[::-1, ::-1, 0, 0])
    ident = np.zeros((3, 3, 2, 2)); ident[1, 1, 0, 0] = ident[1, 1, 1, 1] = 1
    z = np.random.default_rng(0).standard_normal((2, 4, 6, 2))
    np.testing.assert_allclose(N.conv2d(z, ident, None, 1, 1), z)
    # output sizes of the reference geometries: pad 3/k8/s2 halves; pad 1/k4/s2 halves; pad 1/k7/s3; k3/s2 valid
    assert N.conv2d(np.zeros((1, 16, 16, 1)), np.zeros((8, 8, 1, 1)), None, 2, 3).shape[1] == 8
    assert N.conv2d(np.zeros((1, 16, 16, 1)), np.zeros((4, 4, 1, 1)), None, 2, 1).shape[1] == 8
    assert N.conv2d(np.zeros((1, 96, 96, 1)), np.zeros((7, 7, 1, 1)), None, 3, 1).shape[1] == 31
    assert N.conv2d(np.zeros((1, 3, 3, 1)), np.zeros((3, 3, 1, 1)), None, 2, 0).shape[1] == 1
    # strided tap placement: y[i,j] = x[2i+p-1, 2j+q-1]
    x = np.arange(36, dtype=float).reshape(1, 6, 6, 1)
    w = np.zeros((4, 4, 1, 1)); w[3, 0, 0, 0] = 1.0
    y = N.conv2d(x, w, None, 2, 1)
    assert y[0, 1, 1, 0] == x[0, 2 * 1 + 3 - 1, 2 * 1 + 0 - 1, 0]


def test_conv2d_transpose_phase_placement():
    x = np.zeros((1, 2, 2, 1)); x[0, 1, 0, 0] = 1.0
    w = np.arange(1, 5, dtype=float).reshape(2, 2, 1, 1)          # (kh,kw,out,in)
    y = N.conv2d_transpose(x, w, None, 2, 0)
    assert y.shape == (1, 4, 4, 1)
    np.testing.assert_array_equal(y[0, 2:4, 0:2, 0], w[:, :, 0, 0])   # pixel (1,0) -> block rows 2..3, cols 0..1
    assert y.sum() == w.sum()
    # 5x5 'same' stride 1: an impulse at the centre reproduces the kernel itself (no flip)
    x = np.zeros((1, 7, 7, 1)); x[0, 3, 3, 0] = 1.0
    w5 = np.arange(25, dtype=float).reshape(5, 5, 1, 1)
    y = N.conv2d_transpose(x, w5, None, 1, 2)
    assert y.shape == (1, 7, 7, 1)
    np.testing.assert_array_equal(y[0, 1:6, 1:6, 0], w5[:, :, 0, 0])
    # channel roles: kernel axis 2 = outputs, axis 3 = inputs
    w = np.zeros((2, 2, 3, 5)); x = np.ones((1, 2, 2, 5))
    assert N.conv2d_transpose(x, w, np.zeros(3), 2, 0).shape == (1, 4, 4, 3)


def test_bilinear_rows_and_edges():
    x = np.array([0.0, 4.0, 8.0]).reshape(1, 3, 1, 1) * np.ones((1, 3, 2, 1))
    y = N.upsample_bilinear_2x(x)[0, :, 0, 0]
    # half-pixel centres: first/last output copy the edge, interior taps are .25/.75
    np.testing.assert_allclose(y, [0.0, 1.0, 3.0, 5.0, 7.0, 8.0])
    assert N.upsample_bilinear_2x(np.ones((2, 4, 5, 3))).shape == (2, 8, 10, 3)
    np.testing.assert_allclose(N.upsample_bilinear_2x(np.full((1, 3, 3, 1), 2.5)), 2.5)


def test_activations_and_norms():
    np.testing.assert_allclose(N.hard_sigmoid(np.array([-3.0, -2.5, -1.0, 0.0, 1.0, 2.5, 3.0])),
                               [0.0, 0.0, 0.3, 0.5, 0.7, 1.0, 1.0])
    np.testing.assert_allclose(N.leaky_relu(np.array([-2.0, 0.0, 3.0])), [-0.4, 0.0, 3.0])
    # LayerNorm over channels only; constant input -> beta; eps = 1e-3 matters
    x = np.full((1, 2, 2, 4), 7.0)
    np.testing.assert_allclose(N.layer_norm(x, np.ones(4), np.full(4, 0.5)), 0.5)
    r = np.array([[-1.0, 1.0]])
    np.testing.assert_allclose(N.layer_norm(r, np.ones(2), np.zeros(2)), r / np.sqrt(1.0 + 1e-3))
    # BatchNorm: per-channel over all other axes, biased variance
    x = np.stack([np.array([0.0, 2.0, 4.0, 6.0]), np.array([1.0, 1.0, 1.0, 1.0])], -1).reshape(1, 1, 2, 2, 2)
    y, mean, var = N.batch_norm_train(x, np.ones(2), np.zeros(2))
    np.testing.assert_allclose(mean, [3.0, 1.0]); np.testing.assert_allclose(var, [5.0, 0.0])
    np.testing.assert_allclose(y[..., 1], 0.0)
    np.testing.assert_allclose(y[..., 0].ravel(), (np.array([0, 2, 4, 6.0]) - 3) / np.sqrt(5 + 1e-3))


def test_conv_lstm_first_step_and_recurrence():
    # 1x1 spatial, 1 feature: gates are plain affine maps -> closed form
    x = np.array([0.5, -1.0]).reshape(1, 2, 1, 1, 1)
    k = np.zeros((3, 3, 1, 4)); k[1, 1, 0] = [1.0, 2.0, 3.0, 4.0]
    r = np.zeros((3, 3, 1, 4)); r[1, 1, 0] = [0.5, 0.5, 0.5, 0.5]
    b = np.array([0.0, 1.0, 0.0, 0.0])
    h = N.conv_lstm(x, k, r, b)[0, :, 0, 0, 0]
    hs = lambda v: min(max(0.2 * v + 0.5, 0), 1)
    c1 = hs(0.5) * np.tanh(1.5)                                   # f*c0 = 0
    h1 = hs(2.0) * np.tanh(c1)
    z = np.array([-1.0, -2.0 + 1.0, -3.0, -4.0]) + 0.5 * h1
    c2 = hs(z[1]) * c1 + hs(z[0]) * np.tanh(z[2])
    h2 = hs(z[3]) * np.tanh(c2)
    np.testing.assert_allclose(h, [h1, h2], rtol=1e-12)


def test_spectral_norm_and_adam_known_answers():
    # rank-1 matrix: one power iteration recovers sigma exactly
    a, b = np.array([3.0, 4.0]), np.array([1.0, 2.0, 2.0])
    w = np.outer(a, b).reshape(1, 1, 2, 3)
    wn, un, sigma = N.spectral_normalize(w, np.array([[1.0, 0.0, 0.0]]))
    assert abs(sigma - 5.0 * 3.0) < 1e-12
    np.testing.assert_allclose(un, b[None] / 3.0)
    np.testing.assert_allclose(np.linalg.svd(wn.reshape(2, 3), compute_uv=False)[0], 1.0)
    # first Adam step, train.py hyper-parameters, g = 0.01: TF form 3.07e-6 (torch form would be 9.09e-6)
    p, m, v = N.adam_tf(np.zeros(1), np.full(1, 0.01), np.zeros(1), np.zeros(1), 1, 1e-4, 0.5, 0.9, 0.1)
    lr_t = 1e-4 * np.sqrt(0.1) / 0.5
    np.testing.assert_allclose(-p, lr_t * 0.005 / (np.sqrt(1e-5) + 0.1), rtol=1e-12)
    assert abs(-p[0] - 3.07e-6) < 1e-8


def test_philox_known_answer():
    """Philox4x32-10 test vector of Random123 (kat_vectors): counter=0, key=0."""
    r = philox4x32_10(np.zeros(1, dtype=np.uint64), None, 0)[0]
    assert [hex(int(v)) for v in r] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]


# ---------------- (ii) torch restatements agree with the numpy one ------------------------------------
def _t(a):
    return torch.tensor(a, dtype=torch.float64)


@pytest.mark.parametrize("k,s,p,H", [(8, 2, 3, 12), (4, 2, 1, 10), (3, 1, 1, 7), (7, 3, 1, 20), (3, 2, 0, 5)])
def test_torch_conv_matches_numpy(k, s, p, H):
    rng = np.random.default_rng(k)
    x, w, b = rng.standard_normal((2, H, H, 3)), rng.standard_normal((k, k, 3, 4)), rng.standard_normal(4)
    ref = N.conv2d(x, w, b, s, p)
    np.testing.assert_allclose(TM.conv2d(_t(x), _t(w), _t(b), s, p, act=False).numpy(), ref, atol=1e-12)
    ops = TorchOps()
    from oracle.torch_backend import ConvGeom
    y = torch.zeros(ref.shape, dtype=torch.float64)
    ops.conv_fwd(_t(x), ops.pack_weights(_t(w)), _t(b), y, ConvGeom(k, k, s, p))
    np.testing.assert_allclose(y.numpy(), ref, atol=1e-12)


def test_torch_convT_upsample_lstm_norms_match_numpy():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 5, 6, 7))
    w2, w5 = rng.standard_normal((2, 2, 3, 7)), rng.standard_normal((5, 5, 4, 7))
    np.testing.assert_allclose(TM.conv2d_transpose(_t(x), _t(w2), None, 2, 0, act=False).numpy(), N.conv2d_transpose(x, w2, None, 2, 0), atol=1e-12)
    np.testing.assert_allclose(TM.conv2d_transpose(_t(x), _t(w5), None, 1, 2, act=False).numpy(), N.conv2d_transpose(x, w5, None, 1, 2), atol=1e-12)
    np.testing.assert_allclose(TM.upsample_bilinear_2x(_t(x)).numpy(), N.upsample_bilinear_2x(x), atol=1e-13)
    xs = rng.standard_normal((2, 3, 5, 5, 2))
    k, r, b = rng.standard_normal((3, 3, 2, 12)), rng.standard_normal((3, 3, 3, 12)), rng.standard_normal(12)
    np.testing.assert_allclose(TM.conv_lstm(_t(xs), _t(k), _t(r), _t(b)).numpy(), N.conv_lstm(xs, k, r, b), atol=1e-12)
    g, be = rng.uniform(0.5, 1.5, 7), rng.standard_normal(7)
    wd = {"n/gamma": _t(g), "n/beta": _t(be)}
    np.testing.assert_allclose(TM.layer_norm(_t(x), wd, "n").numpy(), N.layer_norm(x, g, be), atol=1e-12)
    wd.update({"n/moving_mean": _t(np.zeros(7)), "n/moving_variance": _t(np.ones(7))})
    np.testing.assert_allclose(TM.batch_norm(_t(x), wd, "n", True, {}).numpy(), N.batch_norm_train(x, g, be)[0], atol=1e-12)
    wn, un = TM.spectral_normalize(_t(w5), _t(rng.standard_normal((1, 7)) * 0.02))
    u0 = rng.standard_normal((1, 7))
    a, b_, _ = N.spectral_normalize(w5, u0)
    wn, un = TM.spectral_normalize(_t(w5), _t(u0))
    np.testing.assert_allclose(wn.numpy(), a, rtol=1e-12); np.testing.assert_allclose(un.numpy(), b_, rtol=1e-12)


def test_torch_generator_matches_numpy_generator():
    from downscaling.engine.networks import GeneratorNet
    from tests.helpers import randomize
    net = GeneratorNet(TorchOps(), 8, 3, 1, 2, 2, feature_channels=32, seed=0)
    w = randomize(net, 5)
    rng = np.random.default_rng(2)
    image, noise = rng.standard_normal((2, 2, 8, 8, 3)), rng.standard_normal((2, 2, 8, 8, 1)) * 0.1
    wn = {k: v.numpy() for k, v in w.items()}
    for training in (False, True):
        ref = N.generator_forward(wn, image, noise, training)
        got = TM.generator_forward(w, _t(image), _t(noise), training, {})
        np.testing.assert_allclose(got.numpy(), ref, atol=1e-10)


def test_wind_speed_weighted_rmse_matches_numpy():
    from downscaling.engine import runtime
    from downscaling.gan.metrics import wind_speed_weighted_rmse
    from oracle.torch_backend import TorchOps
    rng = np.random.default_rng(3)
    a, b = rng.standard_normal((3, 2, 5, 5, 2)) * 5, rng.standard_normal((3, 2, 5, 5, 2)) * 5
    runtime.set_ops(TorchOps(torch.float64))         # the package's metric functions run on the operator backend
    try:
        np.testing.assert_allclose(wind_speed_weighted_rmse(_t(a), _t(b)).numpy(), N.wind_speed_weighted_rmse(a, b), rtol=1e-12)
        # the differentiable form of the reconstruction-loss slot is the same formula
        np.testing.assert_allclose(wind_speed_weighted_rmse(_t(a), _t(b).requires_grad_(True)).detach().numpy(),
                                   N.wind_speed_weighted_rmse(a, b), rtol=1e-12)
    finally:
        runtime.set_ops(None)
