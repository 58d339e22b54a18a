"""Host-logic parity on CPU: the layer engine (explicit forward/backward programs, zero-copy concat
views, time-major layout, fused norm/activation backward, SN/BN state handling, train-step
sequencing) driven by the float64 oracle operator backend must reproduce the independent
autograd restatement of the reference (oracle/torch_model.py) to float64 round-off.

The GPU parity tests (tests/test_model_gpu.py) run the same engine on the HIP backend."""
import numpy as np
import pytest
import torch

from downscaling.engine.networks import DiscriminatorNet, EncoderNet, GeneratorNet, discriminator_plan
from downscaling.engine.trainer import AdamTF, GanEngine, PhiloxSource
from oracle import torch_model as TM
from oracle.torch_backend import TorchOps
from tests.helpers import Draws, grads64, randomize, rel_err, weights64

TOL = 1e-9


@pytest.fixture(scope="module")
def ops():
    return TorchOps()


def _inputs(B, T, S, cin, nz, ch, seed=0):
    g = torch.Generator().manual_seed(seed)
    low = torch.randn(B, T, S, S, cin, generator=g, dtype=torch.float64)
    noise = torch.randn(B, T, S, S, nz, generator=g, dtype=torch.float64) * 0.1
    high = torch.randn(B, T, S, S, ch, generator=g, dtype=torch.float64)
    return low, noise, high


@pytest.mark.parametrize("S,T,training,F", [(12, 2, True, 32), (12, 2, False, 32), (20, 1, True, 32), (16, 3, True, 32),
                                            (12, 2, True, 16), (16, 1, True, 48), (12, 1, False, 80),
                                            (12, 2, True, 24), (16, 1, True, 40), (12, 1, False, 56)])
def test_generator_forward_backward(ops, S, T, training, F):
    """F = 16 / 48 / 80: feature_channels that are multiples of 16 but not of 32 (models.py:20 asserts % 8) — the last
    decoder stage then has feature_channels / 8 = 2 / 6 / 10 channels and runs at the zero-padded width.
    F = 24 / 40 / 56 (8 mod 16): the [conv-transpose path | res_2] concatenation carries zero alignment channels behind its
    feature_channels / 4 = 6 / 10 / 14 wide first segment and the kernel reading it matching zero rows (params.Var.gap)."""
    B, cin, nz, ch = 2, 3, 2, 2
    net = GeneratorNet(ops, S, cin, nz, ch, T, feature_channels=F, seed=3)
    assert (net.c9.w.gap is not None) == (F % 16 == 8)
    w = randomize(net, 11)
    low, noise, _ = _inputs(B, T, S, cin, nz, ch)
    net.set_image(low)
    net.set_noise(noise)
    out_tm = net.forward(B, training)
    out = torch.zeros(B, T, S, S, ch, dtype=torch.float64)
    net.from_time_major(out_tm, out)

    keys = TM.trainable_keys(w)
    for k in keys:
        w[k].requires_grad_(True)
    TM.apply_sn(w, TM.generator_sn_keys(), training)
    st = {}
    ref = TM.generator_forward(w, low, noise, training, st)
    assert rel_err(out, ref) < TOL
    assert float(out_tm[..., ch:].abs().max()) == 0.0
    got = weights64(net)
    for k in TM.generator_sn_keys():          # SN mutated w and u in place (training only)
        assert rel_err(got[k + "/w"], w[k + "/w"]) < TOL and rel_err(got[k + "/sn_u"], w[k + "/sn_u"]) < TOL
    for k, v in st.items():                   # BN moving statistics
        assert rel_err(got[k], v) < TOL
    if not training:
        return
    gout = torch.randn(ref.shape, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    gref = TM._grads((ref * gout).sum(), w, keys)
    dout = ops.zeros(T * B, S, S, 4)
    net.to_time_major(gout, dout)
    net.params.zero_grad()
    net.backward(B, dout)
    g = grads64(net)
    for k in keys:
        assert rel_err(g[k], gref[k]) < 1e-8, k
    # the 4-element alignment slots behind the variables (read as zero pad channels) stay zero
    pads = float(net.params.grads.abs().sum()) - sum(float(v.grad.abs().sum()) for v in net.params.trainable)
    assert abs(pads) < 1e-9 * max(1.0, float(net.params.grads.abs().sum()))
    if net.c9.w.gap is not None:              # the zero rows of the gapped kernel: zero value, zero gradient
        _, pos, width = net.c9.w.gap
        assert float(net.c9.w.value[..., pos:pos + width].abs().max()) == 0.0
        assert float(net.c9.w.grad[..., pos:pos + width].abs().max()) == 0.0


@pytest.mark.parametrize("S,T,Fd,variant", [(12, 2, 8, False), (20, 1, 8, False), (32, 2, 8, False), (24, 1, 8, False),
                                            (20, 1, 16, False), (12, 2, 8, True), (20, 1, 8, True), (32, 2, 8, True)])
def test_discriminator_forward_backward(ops, S, T, Fd, variant):
    """(.., T=1, Fd=16) takes the fused single-timestep ConvLSTM ops (convlstm1_*), the others the general path.
    variant=True: the split connection of models.py:127-130 (shortcut_convolution, tf_utils.py:15-32) as the shipped
    discriminator checkpoint has it — tapped from the concat (S=12, 6x6 stride 7), as one full-size window (S=20) and
    from a 10x10 map with 6x6 windows at stride 12 (S=32)."""
    B, cl, ch = 2, 3, 2
    net = DiscriminatorNet(ops, S, S, cl, ch, T, feature_channels=Fd, seed=4, shortcut_variant=variant)
    assert (net.shortcut is not None) == variant
    w = randomize(net, 12)
    low, _, high = _inputs(B, T, S, cl, 1, ch, seed=1)
    keys = TM.trainable_keys(w)
    for k in keys:
        w[k].requires_grad_(True)
    net.set_low(low)
    high_tm = ops.zeros(T * B, S, S, 4)
    net.to_time_major(high, high_tm)
    net.set_high_tm(high_tm, B)
    score = net.forward(B, training=True).clone()

    TM.apply_sn(w, TM.discriminator_sn_keys(S, variant), True)
    hreq = high.clone().requires_grad_(True)
    ref = TM.discriminator_forward(w, low, hreq, variant)
    assert rel_err(score, ref.reshape(-1)) < TOL
    dscore = torch.randn(B, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    loss = (ref.reshape(-1) * dscore).sum()
    gref = TM._grads(loss, w, keys + [])
    (ghigh,) = torch.autograd.grad((TM.discriminator_forward(w, low, hreq, variant).reshape(-1) * dscore).sum(), hreq)

    net.params.zero_grad()
    dhigh_tm = net.backward(B, dscore.clone(), need_wgrad=True)
    g = grads64(net)
    for k in keys:
        assert rel_err(g[k], gref[k]) < 1e-8, k
    dhigh = torch.zeros_like(high)
    net.from_time_major(dhigh_tm, dhigh)
    assert rel_err(dhigh, ghigh) < 1e-8
    # input-gradient-only pass leaves the parameter gradients untouched
    net.params.zero_grad()
    net.forward(B, training=False)
    d2 = net.backward(B, dscore.clone(), need_wgrad=False).clone()
    assert float(net.params.grads.abs().max()) == 0.0
    assert rel_err(d2, dhigh_tm) < TOL


def test_discriminator_plan_matches_reference_sizes():
    """loop structure at the shipped / benchmark sizes."""
    assert [(b[0], b[5], b[4]) for b in discriminator_plan(256, 32)[0]] == [(7, 84, 64), (7, 27, 128), (7, 8, 256), (7, 2, 512)]
    assert [(b[0], b[5], b[4]) for b in discriminator_plan(128, 32)[0]] == [(7, 42, 64), (7, 13, 128), (7, 3, 256), (3, 1, 512)]
    assert [(b[0], b[5], b[4]) for b in discriminator_plan(96, 32)[0]] == [(7, 31, 64), (7, 9, 128), (7, 2, 256)]
    assert discriminator_plan(256, 32)[1:] == (2, 512) and discriminator_plan(96, 32)[1:] == (2, 256)


@pytest.mark.parametrize("S,T", [(12, 2), (20, 1)])
def test_train_step_matches_oracle(ops, S, T):
    """Two consecutive GAN.train_step calls: weights, optimizer effect, SN/BN state and every
    reported scalar against the autograd restatement consuming the identical Philox stream."""
    B, cin, nz, ch = 2, 3, 2, 2
    gen = GeneratorNet(ops, S, cin, nz, ch, T, feature_channels=32, seed=5)
    disc = DiscriminatorNet(ops, S, S, cin, ch, T, feature_channels=8, seed=6)
    gw, dw = randomize(gen, 21), randomize(disc, 22)
    eng = GanEngine(gen, disc, PhiloxSource(ops, seed=99), noise_std=0.1, n_critic=3)
    g_opt, d_opt = AdamTF(1e-4, 0.5, 0.9, 0.1), AdamTF(4e-4, 0.5, 0.9, 0.1)
    og, od = TM.AdamTF(1e-4), TM.AdamTF(4e-4)
    draws = Draws(eng.noise.seed, B, T, S, nz, ch, 0.1)
    for step in range(2):
        low, _, high = _inputs(B, T, S, cin, nz, ch, seed=30 + step)
        res = eng.train_step(low, high, g_opt, d_opt)
        ref = TM.train_step(gw, dw, low, high, draws, og, od)
        for k in ("g_loss", "g_disc_loss", "d_loss", "d_gradient_pen", "g_gradient_param", "d_gradient_param", "_d_loss_train"):
            assert rel_err(res[k], ref[k]) < 1e-7, (step, k)
        for net, w in ((gen, gw), (disc, dw)):
            got = weights64(net)
            for k in w:
                assert rel_err(got[k], w[k]) < 1e-7, (step, k)


def _relativistic(real_output, fake_output):
    """A loss that COUPLES the two score vectors (relativistic average hinge): d loss / d real depends on fake."""
    return (torch.relu(1.0 - (real_output - fake_output.mean())).mean() + torch.relu(1.0 + (fake_output - real_output.mean())).mean())


def _wasserstein_again(real_output, fake_output):
    return fake_output.mean() - real_output.mean()


def _per_sample_hinge(real_output, fake_output):
    """A compiled loss that returns one value PER SAMPLE ([B]): Keras weights it elementwise and reduces SUM_OVER_BATCH_SIZE."""
    return (torch.relu(1.0 - real_output) + torch.relu(1.0 + fake_output)).reshape(-1)


def test_per_sample_custom_loss_is_weighted_elementwise(ops):
    """a per-sample loss vector with non-uniform sample weights is sum(loss_b * sw_b) / B (compute_weighted_loss), not
    loss * mean(sw); without weights it is the mean.  Engine (coupled critic path) against the autograd restatement."""
    B, cin, nz, ch, S, T = 2, 3, 2, 2, 12, 1
    for sw in (torch.tensor([0.25, 1.75], dtype=torch.float64), None):
        gen = GeneratorNet(ops, S, cin, nz, ch, T, feature_channels=32, seed=5)
        disc = DiscriminatorNet(ops, S, S, cin, ch, T, feature_channels=8, seed=6)
        gw, dw = randomize(gen, 21), randomize(disc, 22)
        eng = GanEngine(gen, disc, PhiloxSource(ops, seed=99), noise_std=0.1, n_critic=1)
        draws = Draws(eng.noise.seed, B, T, S, nz, ch, 0.1)
        low, _, high = _inputs(B, T, S, cin, nz, ch, seed=40)
        res = eng.train_step(low, high, AdamTF(1e-4, 0.5, 0.9, 0.1), AdamTF(4e-4, 0.5, 0.9, 0.1), d_loss_fn=_per_sample_hinge, sample_weight=sw)
        ref = TM.train_step(gw, dw, low, high, draws, TM.AdamTF(1e-4), TM.AdamTF(4e-4), n_critic=1, d_loss_fn=_per_sample_hinge, sample_weight=sw)
        for k in ("d_loss", "_d_loss_train", "d_gradient_param"):
            assert rel_err(res[k], ref[k]) < 1e-7, (k, sw)
        got = weights64(disc)
        for k in dw:
            assert rel_err(got[k], dw[k]) < 1e-7, (k, sw)


@pytest.mark.parametrize("loss", [_wasserstein_again, _relativistic], ids=["wasserstein_as_custom", "relativistic_hinge"])
def test_train_step_with_custom_discriminator_loss(ops, loss):
    """GAN.compile(discriminator_loss=<any callable>) (ganbase.py:44-45 calls compiled_loss(real_output, fake_output)): the
    coupled critic path against the autograd restatement; the Wasserstein form written as a custom callable must also
    reproduce the built-in path."""
    B, cin, nz, ch, S, T = 2, 3, 2, 2, 12, 2
    gen = GeneratorNet(ops, S, cin, nz, ch, T, feature_channels=32, seed=5)
    disc = DiscriminatorNet(ops, S, S, cin, ch, T, feature_channels=8, seed=6)
    gw, dw = randomize(gen, 21), randomize(disc, 22)
    eng = GanEngine(gen, disc, PhiloxSource(ops, seed=99), noise_std=0.1, n_critic=2)
    g_opt, d_opt = AdamTF(1e-4, 0.5, 0.9, 0.1), AdamTF(4e-4, 0.5, 0.9, 0.1)
    og, od = TM.AdamTF(1e-4), TM.AdamTF(4e-4)
    draws = Draws(eng.noise.seed, B, T, S, nz, ch, 0.1)
    for step in range(2):
        low, _, high = _inputs(B, T, S, cin, nz, ch, seed=40 + step)
        res = eng.train_step(low, high, g_opt, d_opt, d_loss_fn=loss)
        ref = TM.train_step(gw, dw, low, high, draws, og, od, n_critic=2, d_loss_fn=loss)
        for k in ("g_loss", "d_loss", "d_gradient_pen", "d_gradient_param", "_d_loss_train"):
            assert rel_err(res[k], ref[k]) < 1e-7, (step, k)
        for net, w in ((gen, gw), (disc, dw)):
            got = weights64(net)
            for k in w:
                assert rel_err(got[k], w[k]) < 1e-7, (step, k)
    if loss is _wasserstein_again:
        gen2 = GeneratorNet(ops, S, cin, nz, ch, T, feature_channels=32, seed=5)
        disc2 = DiscriminatorNet(ops, S, S, cin, ch, T, feature_channels=8, seed=6)
        randomize(gen2, 21), randomize(disc2, 22)
        eng2 = GanEngine(gen2, disc2, PhiloxSource(ops, seed=99), noise_std=0.1, n_critic=2)
        g2, d2 = AdamTF(1e-4, 0.5, 0.9, 0.1), AdamTF(4e-4, 0.5, 0.9, 0.1)
        for step in range(2):
            low, _, high = _inputs(B, T, S, cin, nz, ch, seed=40 + step)
            eng2.train_step(low, high, g2, d2)
        a, b = weights64(disc), weights64(disc2)
        for k in a:
            assert rel_err(a[k], b[k]) < 1e-12, k


def _ws_rmse_engine(real_output, fake_output):
    """The package's metrics.wind_speed_weighted_rmse (metrics.py:32-45) in the reconstruction-loss slot: differentiable
    when its second argument requires a gradient."""
    from downscaling.gan.metrics import wind_speed_weighted_rmse
    return wind_speed_weighted_rmse(real_output, fake_output)


@pytest.mark.parametrize("case", ["encoder", "ws_rmse_vector", "sample_weight", "shortcut"])
def test_train_step_remaining_branches(ops, case):
    """The branches of GAN.train_step the default call does not take (VERDICT r4 missing #2), engine vs the autograd
    restatement, two steps each:
      encoder         reconstruction_loss(encoder, 0.5) in the generator step (ganbase.py:57-59, train.py:19-26) with the
                      reference's own feature extractor (EncoderNet, autoencoder.py:23-36);
      ws_rmse_vector  wind_speed_weighted_rmse (metrics.py:32-45) as content loss: a per-sample VECTOR, so gen_loss is a
                      vector and the tape differentiates its sum;
      sample_weight   the third element of the data tuple (ganbase.py:23,44,67);
      shortcut        the discriminator graph of the shipped checkpoint inside a train step (lazily zeroed gradients)."""
    B, cin, nz, ch, S, T = 2, 3, 2, 2, 24 if case == "encoder" else 12, 2
    variant = case == "shortcut"
    gen = GeneratorNet(ops, S, cin, nz, ch, T, feature_channels=32, seed=5)
    disc = DiscriminatorNet(ops, S, S, cin, ch, T, feature_channels=8, seed=6, shortcut_variant=variant)
    gw, dw = randomize(gen, 21), randomize(disc, 22)
    eng = GanEngine(gen, disc, PhiloxSource(ops, seed=99), noise_std=0.1, n_critic=2)
    g_opt, d_opt = AdamTF(1e-4, 0.5, 0.9, 0.1), AdamTF(4e-4, 0.5, 0.9, 0.1)
    og, od = TM.AdamTF(1e-4), TM.AdamTF(4e-4)
    draws = Draws(eng.noise.seed, B, T, S, nz, ch, 0.1)
    kw_e, kw_r = {}, {}
    if case == "encoder":
        from downscaling.engine.networks import EncoderNet
        from downscaling.gan.train import reconstruction_loss
        from downscaling.autoencoder.autoencoder import Encoder
        enc = EncoderNet(ops, S, T, 4, seed=7)
        ew = randomize(enc, 31)
        kw_e["reconstruction_loss"] = reconstruction_loss(Encoder(enc), 0.5)
        kw_r["reconstruction_loss"] = TM.reconstruction_loss(lambda x: TM.encoder_forward(ew, x, 4), 0.5)
    elif case == "ws_rmse_vector":
        kw_e["reconstruction_loss"] = _ws_rmse_engine
        kw_r["reconstruction_loss"] = TM.wind_speed_weighted_rmse
    elif case == "sample_weight":
        kw_e["sample_weight"] = kw_r["sample_weight"] = torch.tensor([0.25, 1.75], dtype=torch.float64) * 0.8
    for step in range(2):
        low, _, high = _inputs(B, T, S, cin, nz, ch, seed=60 + step)
        res = eng.train_step(low, high, g_opt, d_opt, **kw_e)
        ref = TM.train_step(gw, dw, low, high, draws, og, od, n_critic=2, shortcut_variant=variant, **kw_r)
        keys = ["g_loss", "g_disc_loss", "d_loss", "d_gradient_pen", "g_gradient_param", "d_gradient_param", "_d_loss_train"]
        if "reconstruction_loss" in kw_e:
            keys.append("g_reco_loss")
        for k in keys:
            assert rel_err(res[k], ref[k]) < 1e-7, (step, k)
        for net, w in ((gen, gw), (disc, dw)):
            got = weights64(net)
            for k in w:
                assert rel_err(got[k], w[k]) < 1e-7, (step, k)
    # test_step (ganbase.py:96-113) on the trained pair
    low, _, high = _inputs(B, T, S, cin, nz, ch, seed=69)
    t = eng.test_step(low, high)
    tr = TM.test_step(gw, dw, low, high, draws, shortcut_variant=variant)
    assert rel_err(t["loss"], tr["loss"]) < 1e-7
    fake = torch.zeros(B, T, S, S, ch, dtype=torch.float64)
    gen.from_time_major(eng.last_fake_tm, fake)
    assert rel_err(fake, tr["generated"]) < 1e-7


@pytest.mark.parametrize("S,T,latent", [(96, 2, 96), (24, 1, 4), (40, 2, 8)])
def test_encoder_forward_and_input_gradient(ops, S, T, latent):
    """AutoEncoder.make_encoder (autoencoder/autoencoder.py:23-36), the reconstruction-loss feature extractor: forward
    and d/dx against the autograd restatement.  (96, ., 96) is the reference's own build (features_encoding.py:10-13:
    three 5x5 stride-3 blocks, 144 -> 96 in one Dense); (24, ., 4) and (40, ., 8) take the two-Dense branch."""
    B = 2
    net = EncoderNet(ops, S, T, latent, seed=7)
    w = randomize(net, 31)
    x = torch.randn(B, T, S, S, 2, generator=torch.Generator().manual_seed(8), dtype=torch.float64)
    got = net.forward(x).clone()
    xr = x.clone().requires_grad_(True)
    ref = TM.encoder_forward(w, xr, latent)
    assert got.shape == ref.shape == (B, T, latent)
    assert rel_err(got, ref) < 1e-10
    g = torch.randn(ref.shape, generator=torch.Generator().manual_seed(9), dtype=torch.float64)
    (gref,) = torch.autograd.grad((ref * g).sum(), xr)
    assert rel_err(net.backward_input(g), gref) < 1e-9


@pytest.mark.parametrize("variant", [False, True])
def test_lazy_zero_grad_equals_eager(ops, monkeypatch, variant):
    """variant=True: the shortcut kernel of the shipped checkpoint's graph, whose gradient is written by _shortcut_bwd and not
    by Conv.backward_weights (it accumulated onto the stale slot and settle() then zeroed it).
    ParamStore.zero_grad(lazy=True) (the trainer's critic updates): big convolution kernels are not zero-filled but
    marked fresh, their first weight-gradient launch stores instead of accumulating, a second pass accumulates, and
    settle() zero-fills what no pass wrote — the gradient buffer must equal the eagerly zeroed one in every case."""
    from downscaling.engine.params import ParamStore
    monkeypatch.setattr(ParamStore, "LAZY_MIN", 1)          # (every Conv kernel of the small test network takes part)
    B, S, T, cl, ch = 2, 20, 1, 3, 2
    net = DiscriminatorNet(ops, S, S, cl, ch, T, feature_channels=8, seed=4, shortcut_variant=variant)
    assert (net.shortcut is not None) == variant
    randomize(net, 12)
    low, _, high = _inputs(B, T, S, cl, 1, ch, seed=1)
    net.set_low(low)
    high_tm = ops.zeros(T * B, S, S, 4)
    net.to_time_major(high, high_tm)
    net.set_high_tm(high_tm, B)
    dscore = torch.randn(B, generator=torch.Generator().manual_seed(2), dtype=torch.float64)

    def passes(lazy, n):
        net.params.grads.fill_(123.0)                       # garbage a lazy zeroing must never let through
        net.params.zero_grad(lazy=lazy)
        assert any(v.fresh for v in net.params.trainable) == lazy
        for _ in range(n):
            net.forward(B, training=False)
            net.backward(B, dscore.clone(), need_wgrad=True, need_input_grad=False)
        net.params.settle()
        assert not any(v.fresh for v in net.params.trainable)
        return net.params.grads.clone()

    for n in (0, 1, 2):      # 0: nothing written -> settle() zero-fills; 1: stored; 2: stored, then accumulated
        eager, lazy = passes(False, n), passes(True, n)
        assert float((eager - lazy).abs().max()) == 0.0, n
    assert float(passes(True, 1).abs().max()) > 0.0
    if variant:
        net.params.zero_grad(lazy=True)
        assert net.shortcut["conv"].w.fresh
        net.forward(B, training=False)
        net.backward(B, dscore.clone(), need_wgrad=True, need_input_grad=False)
        net.params.settle()
        assert float(net.shortcut["conv"].w.grad.abs().max()) > 0.0
