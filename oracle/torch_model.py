"""TEST INFRASTRUCTURE — CPU restatement of the reference models and GAN train step.

A second, independent statement of the path (it shares no code with the layer engine or with
oracle/torch_backend.py): the generator / discriminator graphs and `GAN.train_step` written as plain
differentiable torch-CPU functions in float64, gradients by torch.autograd.  It follows

  /root/reference/src/downscaling/gan/models.py:9-73    make_generator
  /root/reference/src/downscaling/gan/models.py:76-142  make_discriminator
  /root/reference/src/downscaling/tf_utils.py:7-12      img_size / channels
  /root/reference/src/downscaling/gan/ganbase.py:21-94  GAN.train_step (incl. sample_weight :23,44,67 and the
                                                         reconstruction-loss slot :57-59)
  /root/reference/src/downscaling/gan/ganbase.py:96-113 GAN.test_step
  /root/reference/src/downscaling/gan/train.py:11-12,19-26,34-35,57-58   losses, reconstruction_loss, Adam hyper-parameters
  /root/reference/src/downscaling/gan/metrics.py:32-45  wind_speed_weighted_rmse (usable as content loss)

with the TensorFlow 2.4.3 / tensorflow-addons 0.14.0 layer semantics restated from their published
definitions (those packages are third-party dependencies pinned in requirements.txt:2-3, absent from
/root/reference and from this image):
  Conv2D = cross-correlation on NHWC with HWIO kernels; Conv2DTranspose kernel is (kh,kw,out,in);
  LeakyReLU(0.2) applied after the bias; BatchNormalization eps 1e-3, momentum .99, biased batch
  variance; LayerNormalization over the last axis, eps 1e-3; ConvLSTM2D gate order i,f,c,o with
  hard_sigmoid = clip(0.2x+0.5,0,1) recurrent activation; UpSampling2D bilinear = half-pixel centres
  with edge clamp; tfa SpectralNormalization = one power iteration, w <- w/sigma in place, no
  gradient through sigma; Adam with epsilon outside the bias correction.

PARITY UNPINNED: the reference has no tests / golden vectors and TensorFlow cannot run here, so this
restatement is pinned only by hand-computable known-answer tests (tests/test_oracle_known_answers.py).

Weights are passed as {TF checkpoint key: tensor in TF shape}; activations are [B,T,H,W,C].
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch
import torch.nn.functional as F

LRELU, BN_EPS, BN_MOM, LN_EPS = 0.2, 1e-3, 0.99, 1e-3
L = "layer_with_weights-"


def _nchw(x):   # [M,H,W,C] -> [M,C,H,W]
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


def _td(x):     # TimeDistributed: fold time into the batch
    B, T = x.shape[:2]
    return x.reshape(B * T, *x.shape[2:]), (B, T)


def _untd(x, bt):
    return x.reshape(*bt, *x.shape[1:])


def conv2d(x, w_hwio, b, stride=1, pad=0, act=True):
    y = F.conv2d(_nchw(x), w_hwio.permute(3, 2, 0, 1), b, stride=stride, padding=pad)
    y = _nhwc(y)
    return F.leaky_relu(y, LRELU) if act else y


def conv2d_transpose(x, w_hwoi, b, stride, crop, act=True):
    """Keras Conv2DTranspose, kernel (kh,kw,out,in): y[n,i*s+p-crop,j*s+q-crop,o] += x[n,i,j,c]*w[p,q,o,c]."""
    y = F.conv_transpose2d(_nchw(x), w_hwoi.permute(3, 2, 0, 1), b, stride=stride, padding=crop)
    y = _nhwc(y)
    return F.leaky_relu(y, LRELU) if act else y


def hard_sigmoid(x):
    return torch.clamp(0.2 * x + 0.5, 0.0, 1.0)


def conv_lstm(x, kernel, rec, bias):
    """x [B,T,H,W,C] -> all h_t [B,T,H,W,F]."""
    B, T = x.shape[:2]
    Fh = rec.shape[2]
    h = torch.zeros(B, x.shape[2], x.shape[3], Fh, dtype=x.dtype)
    c = torch.zeros_like(h)
    outs = []
    for t in range(T):
        z = conv2d(x[:, t], kernel, bias, 1, 1, act=False) + conv2d(h, rec, None, 1, 1, act=False)
        zi, zf, zc, zo = torch.split(z, Fh, dim=-1)
        i, f, o = hard_sigmoid(zi), hard_sigmoid(zf), hard_sigmoid(zo)
        c = f * c + i * torch.tanh(zc)
        h = o * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, 1)


def batch_norm(x, w, name, training, new_state):
    """x [B,T,H,W,C]; statistics over every axis but the last."""
    gamma, beta = w[name + "/gamma"], w[name + "/beta"]
    if training:
        dims = tuple(range(x.dim() - 1))
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
        if new_state is not None:
            new_state[name + "/moving_mean"] = (w[name + "/moving_mean"] * BN_MOM + mean * (1 - BN_MOM)).detach()
            # TF 2.4: 5-D input takes the fused op, whose batch_variance (averaged into moving_variance) is the
            # Bessel-corrected one; the normalisation below uses the biased variance
            n = x.numel() // x.shape[-1]
            new_state[name + "/moving_variance"] = (w[name + "/moving_variance"] * BN_MOM
                                                    + x.var(dims, unbiased=n > 1) * (1 - BN_MOM)).detach()
    else:
        mean, var = w[name + "/moving_mean"], w[name + "/moving_variance"]
    return (x - mean) / torch.sqrt(var + BN_EPS) * gamma + beta


def layer_norm(x, w, name):
    return F.layer_norm(x, (x.shape[-1],), w[name + "/gamma"], w[name + "/beta"], LN_EPS)


def upsample_bilinear_2x(x):
    return _nhwc(F.interpolate(_nchw(x), scale_factor=2, mode="bilinear", align_corners=False))


def spectral_normalize(w, u):
    """tfa SpectralNormalization.normalize_weights (power_iterations=1).  Returns (w/sigma, u')."""
    with torch.no_grad():
        W = w.reshape(-1, w.shape[-1])
        def l2n(v):
            return v / torch.sqrt(torch.clamp((v * v).sum(), min=1e-12))
        v = l2n(u @ W.t())
        un = l2n(v @ W)
        sigma = (v @ W @ un.t()).reshape(())
    return w / sigma, un


def apply_sn(w, keys, training):
    """In-place SN update of every `.../w` + `.../sn_u` pair in `keys` order (training only)."""
    if not training:
        return
    for k in keys:
        wn, un = spectral_normalize(w[k + "/w"].detach(), w[k + "/sn_u"])
        with torch.no_grad():
            w[k + "/w"].copy_(wn)
            w[k + "/sn_u"].copy_(un)


def generator_sn_keys():
    return [L + f"{i}/layer" for i in (0, 2, 5, 7)]


def generator_forward(w, image, noise, training=False, new_state=None):
    """make_generator graph, models.py:28-72.  SN updates must be applied by the caller (apply_sn)."""
    x = torch.cat([image, noise], -1)                                                  # :28
    xf, bt = _td(x)
    xf = conv2d(xf, w[L + "0/layer/w"], w[L + "0/layer/layer/bias"], 2, 3)             # :32-33 (pad 3, 8x8 s2)
    x = batch_norm(_untd(xf, bt), w, L + "1", training, new_state)                     # :34
    res_2 = x
    xf, _ = _td(x)
    xf = conv2d(xf, w[L + "2/layer/w"], w[L + "2/layer/layer/bias"], 2, 1)             # :38-39
    x = batch_norm(_untd(xf, bt), w, L + "3", training, new_state)                     # :40
    res_4 = x
    x = conv_lstm(x, w[L + "4/cell/kernel"], w[L + "4/cell/recurrent_kernel"], w[L + "4/cell/bias"])   # :45
    xf, _ = _td(x)
    xf = conv2d(xf, w[L + "5/layer/w"], w[L + "5/layer/layer/bias"], 1, 1)             # :49
    x = batch_norm(_untd(xf, bt), w, L + "6", training, new_state)                     # :50
    x = torch.cat([x, res_4], -1)                                                      # :54
    xf, _ = _td(x)
    xf = conv2d_transpose(xf, w[L + "7/layer/w"], w[L + "7/layer/layer/bias"], 2, 0)   # :55
    x = batch_norm(_untd(xf, bt), w, L + "8", training, new_state)                     # :56
    x = torch.cat([x, res_2], -1)                                                      # :60
    xf, _ = _td(x)
    xf = upsample_bilinear_2x(xf)                                                      # :62
    xf = conv2d_transpose(xf, w[L + "9/layer/kernel"], w[L + "9/layer/bias"], 1, 2)    # :63-64 (5x5 'same')
    x = batch_norm(_untd(xf, bt), w, L + "10", training, new_state)                    # :69
    xf, _ = _td(x)
    xf = conv2d(xf, w[L + "11/layer/kernel"], w[L + "11/layer/bias"], 1, 1, act=False)  # :70-71
    return _untd(xf, bt)


def shortcut_geometry(src_size, target):
    """kernel / stride / padding of shortcut_convolution (tf_utils.py:15-32) from a src_size map to a target map."""
    if target == 1:
        return src_size, 1, 0                                                           # :18-21: one full-size valid conv
    stride = math.ceil((2 + src_size) / (target - 1))                                  # :23
    pad = math.ceil((stride * (target - 1) - src_size) / 2) + 1 + 2                    # :24-25
    return stride * (1 - target) + src_size + 2 * pad, stride, pad                     # :26


def discriminator_layout(size, channels, shortcut_variant=False):
    """Loop structure of models.py:111-136 -> ([(conv key index, LN key index, k, stride, pad)], final index, shortcut).
    shortcut_variant=True is the graph the shipped weights-55 discriminator was trained with (SURVEY 8 a2 note 2): the
    split connection of models.py:127-130 taken when the `>= 4` loop ran once (the published test `i > 1` can never
    succeed).  shortcut = None or dict(src=number of blocks before the tap, conv=key index, ln=key index, k, s, p)."""
    blocks, idx = [], 6
    while size >= 16:
        blocks.append((idx, idx + 1, 7, 3, 1)); size = (size + 2 - 7) // 3 + 1; idx += 2
    n_src, src_size = len(blocks), size
    i = 0
    while size >= 4:
        blocks.append((idx, idx + 1, 7, 3, 1)); size = (size + 2 - 7) // 3 + 1; idx += 2; i += 1
    assert i <= 1, "two passes of the `>= 4` loop cannot happen (the second conv would have zero output)"
    shortcut = None
    if shortcut_variant and i == 1:
        # checkpoint numbering: conv_<size> idx-2, shortcut_conv idx-1, LN(main) idx, LN(shortcut) idx+1
        ci, _, k, s, p = blocks[-1]
        blocks[-1] = (ci, ci + 2, k, s, p)
        sk, ss, sp = shortcut_geometry(src_size, size)
        shortcut = dict(src=n_src, conv=ci + 1, ln=ci + 3, k=sk, s=ss, p=sp)
        idx = ci + 4
    while size > 2:
        blocks.append((idx, idx + 1, 3, 2, 0)); size = (size - 3) // 2 + 1; idx += 2
    return blocks, idx, shortcut


def discriminator_sn_keys(size, shortcut_variant=False):
    blocks, _, sc = discriminator_layout(size, 0, shortcut_variant)
    return [L + "2/layer", L + "3/layer"] + [L + f"{b[0]}/layer" for b in blocks] + ([L + f"{sc['conv']}/layer"] if sc else [])


def discriminator_forward(w, low, high, shortcut_variant=False):
    """make_discriminator graph, models.py:93-140 -> scores [B,1].  (LayerNorm has no train/infer split.)"""
    hr = conv_lstm(high, w[L + "0/cell/kernel"], w[L + "0/cell/recurrent_kernel"], w[L + "0/cell/bias"])   # :93
    hf, bt = _td(hr)
    hf = conv2d(hf, w[L + "2/layer/w"], w[L + "2/layer/layer/bias"], 1, 1)             # :94-96
    hr = layer_norm(_untd(hf, bt), w, L + "4")                                         # :97
    mix = torch.cat([low, high], -1)                                                   # :100
    mix = conv_lstm(mix, w[L + "1/cell/kernel"], w[L + "1/cell/recurrent_kernel"], w[L + "1/cell/bias"])   # :101
    mf, _ = _td(mix)
    mf = conv2d(mf, w[L + "3/layer/w"], w[L + "3/layer/layer/bias"], 1, 1)             # :102-104
    mix = layer_norm(_untd(mf, bt), w, L + "5")                                        # :105
    x = torch.cat([hr, mix], -1)                                                       # :108
    blocks, idx, sc = discriminator_layout(x.shape[2], x.shape[-1], shortcut_variant)
    for n, (i, j, k, s, p) in enumerate(blocks):                                       # :111-136
        if sc and n == sc["src"]:
            shortcut = x                                                               # :118
        xf, _ = _td(x)
        xf = conv2d(xf, w[L + f"{i}/layer/w"], w[L + f"{i}/layer/layer/bias"], s, p)
        x = layer_norm(_untd(xf, bt), w, L + f"{j}")
        if sc and n == sc["src"]:                                                      # :127-130, tf_utils.py:15-32
            sf, _ = _td(shortcut)
            sf = conv2d(sf, w[L + f"{sc['conv']}/layer/w"], w[L + f"{sc['conv']}/layer/layer/bias"], sc["s"], sc["p"])
            x = x + layer_norm(_untd(sf, bt), w, L + f"{sc['ln']}")
    xf, _ = _td(x)
    xf = xf.reshape(xf.shape[0], -1)                                                   # :137 Flatten (H,W,C)
    s = xf @ w[L + f"{idx}/layer/kernel"] + w[L + f"{idx}/layer/bias"]                 # :138 Dense(1)
    return _untd(s, bt).mean(1)                                                        # :139 GlobalAveragePooling1D


def encoder_forward(w, x, latent):
    """AutoEncoder.make_encoder (autoencoder/autoencoder.py:23-36), inference mode -> [B,T,latent]."""
    idx = 0
    bt = None
    while x.shape[2] >= 7:                                                             # :26
        xf, bt = _td(x)
        xf = conv2d(xf, w[L + f"{idx}/layer/w"], w[L + f"{idx}/layer/layer/bias"], 3, 1)   # :27-29 (ZeroPadding2D(1))
        x = layer_norm(_untd(xf, bt), w, L + f"{idx + 1}")                             # :30
        idx += 2
    xf, bt = _td(x)
    xf = xf.reshape(xf.shape[0], -1)                                                   # :31 Flatten (H,W,C)
    if xf.shape[-1] > 2 * latent:                                                      # :32-34
        xf = xf @ w[L + f"{idx}/layer/kernel"] + w[L + f"{idx}/layer/bias"]
        idx += 1
    xf = xf @ w[L + f"{idx}/layer/kernel"] + w[L + f"{idx}/layer/bias"]                # :35
    return _untd(xf, bt)


class AdamTF:
    def __init__(self, lr, b1=0.5, b2=0.9, eps=0.1):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, b1, b2, eps, 0
        self.m, self.v = {}, {}

    def step(self, w, grads):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        with torch.no_grad():
            for k, g in grads.items():
                m = self.m.setdefault(k, torch.zeros_like(g))
                v = self.v.setdefault(k, torch.zeros_like(g))
                m += (1 - self.b1) * (g - m)
                v += (1 - self.b2) * (g * g - v)
                w[k] -= lr_t * m / (torch.sqrt(v) + self.eps)


def trainable_keys(w):
    return [k for k in w if not (k.endswith("sn_u") or "moving_" in k)]


def _grads(loss, w, keys):
    gs = torch.autograd.grad(loss, [w[k] for k in keys], allow_unused=True)
    return {k: (g if g is not None else torch.zeros_like(w[k])) for k, g in zip(keys, gs)}


def weighted_loss(value, sample_weight):
    """What Keras' `compiled_loss(y_true, y_pred, sample_weight)` makes of a plain-function loss (TF 2.4.3
    `LossesContainer` -> `LossFunctionWrapper(fn, reduction=AUTO)` -> `compute_weighted_loss`): the function's value
    (a scalar for train.py:11-12) times the per-sample weights, reduced SUM_OVER_BATCH_SIZE — sum(value * sw) / numel.
    No weights: the value itself.  ganbase.py:44,67."""
    if sample_weight is None:
        return value if value.numel() <= 1 else value.reshape(-1).sum() / value.numel()
    sw = torch.as_tensor(sample_weight, dtype=value.dtype).reshape(-1)
    w = value.reshape(-1) * sw if value.numel() > 1 else value * sw
    return w.sum() / w.numel()


def wind_speed_weighted_rmse(real_output, fake_output):
    """gan/metrics.py:32-45 -> [B]."""
    u, v = real_output[..., 0], real_output[..., 1]
    u_hat, v_hat = fake_output[..., 0], fake_output[..., 1]
    est = torch.sqrt(u_hat ** 2 + v_hat ** 2)
    rea = torch.sqrt(u ** 2 + v ** 2)
    epsilon, t = 4, 0.425
    beta = (epsilon + rea) / (epsilon + est)
    tau = torch.where(est >= rea, torch.full_like(est, t), torch.full_like(est, 1 - t))
    result = tau * ((u_hat - beta * u) ** 2 + (v_hat - beta * v) ** 2)
    result = torch.where(torch.isnan(result), torch.zeros_like(result), result)
    return torch.sqrt(result.mean(dim=(1, 2, 3)))


class reconstruction_loss:
    """gan/train.py:19-26."""

    def __init__(self, feature_extractor, coefficient=1.0):
        self.feature_extractor, self.coefficient = feature_extractor, coefficient

    def __call__(self, low_res, high_res):
        delta = self.feature_extractor(low_res) - self.feature_extractor(high_res)
        return self.coefficient * torch.mean(torch.sqrt(torch.sum(delta ** 2, dim=-1)))


def train_step(gw, dw, low, high, draws, g_opt, d_opt, n_critic=3, gamma=100.0, d_loss_fn=None, sample_weight=None,
               reconstruction_loss=None, shortcut_variant=False):
    """GAN.train_step, ganbase.py:21-94.  `draws` supplies the random tensors in call order:
    draws.noise() [B,T,S,S,nz], draws.eps() [B], draws.inst() [B,T,S,S,ch].  Mutates gw/dw in place.
    d_loss_fn(real_output [B,1], fake_output [B,1]) -> scalar: the discriminator's compiled loss
    (ganbase.py:44-45; default the Wasserstein form of train.py:11-12).
    sample_weight [B] (ganbase.py:23): third element of the data tuple, weights the compiled discriminator loss (:44, :67).
    reconstruction_loss(low_res[..., :2], fake_high_res) (ganbase.py:57-59): added to the generator's loss; a non-scalar
    value makes gen_loss non-scalar and `tape.gradient` differentiates the SUM of its elements (TF semantics).
    shortcut_variant: the discriminator graph of the shipped checkpoint (discriminator_layout)."""
    S = high.shape[2]
    gk, dk = trainable_keys(gw), trainable_keys(dw)
    for k in gk:
        gw[k].requires_grad_(True)
    for k in dk:
        dw[k].requires_grad_(True)
    g_sn, d_sn = generator_sn_keys(), discriminator_sn_keys(S, shortcut_variant)
    sv = shortcut_variant
    B = low.shape[0]
    wasserstein = (lambda r, f: -(r.mean() - f.mean()))                                # train.py:11-12
    loss_fn = wasserstein if d_loss_fn is None else d_loss_fn
    for _ in range(n_critic):                                                          # :26
        noise = draws.noise()                                                          # :28
        apply_sn(gw, g_sn, True)
        st = {}
        with torch.no_grad():
            fake = generator_forward(gw, low, noise, True, st)                         # :29
        for k, v in st.items():
            gw[k].copy_(v)
        eps = draws.eps().reshape(B, 1, 1, 1, 1)                                       # :30
        combined = (eps * high + (1 - eps) * fake).requires_grad_(True)                # :31
        apply_sn(dw, d_sn, True)
        out = discriminator_forward(dw, low, combined, sv)                             # :32-34
        (gimg,) = torch.autograd.grad(out.sum(), combined)                             # :35
        gnorm = torch.sqrt((gimg ** 2).sum((1, 2, 3)))                                 # :36
        gradient_reg = gamma * ((gnorm - 1) ** 2).mean()                               # :37
        hr = high + draws.inst()                                                       # :40
        apply_sn(dw, d_sn, True)
        # the real pass keeps the variable values it read (a copy: the generated pass' SN update overwrites them in place;
        # TF reads the variable value at forward time), the tape sums both passes' partial derivatives per variable
        w_real = {k: (v.detach().clone().requires_grad_(True) if k in dk else v.detach().clone()) for k, v in dw.items()}
        real_s = discriminator_forward(w_real, low, hr, sv)                            # :41
        fhr = fake + draws.inst()                                                      # :42
        apply_sn(dw, d_sn, True)
        fake_s = discriminator_forward(dw, low, fhr, sv)                               # :43
        loss = weighted_loss(loss_fn(real_s, fake_s), sample_weight)                   # :44 compiled_loss(real, fake, sw, ...)
        both = torch.autograd.grad(loss, [w_real[k] for k in dk] + [dw[k] for k in dk], allow_unused=True)
        zero = lambda g, k: g if g is not None else torch.zeros_like(dw[k])   # noqa: E731
        g_real = {k: zero(g, k) for k, g in zip(dk, both[:len(dk)])}
        g_fake = {k: zero(g, k) for k, g in zip(dk, both[len(dk):])}
        disc_loss = loss.detach() + gradient_reg.detach()                              # :45 regularization_losses
        d_grads = {k: g_real[k] + g_fake[k] for k in dk}                               # :46
        d_opt.step(dw, d_grads)                                                        # :47
    noise = draws.noise()                                                              # :51
    apply_sn(gw, g_sn, True)
    st = {}
    fake = generator_forward(gw, low, noise, True, st)                                 # :52
    apply_sn(dw, d_sn, True)
    score = discriminator_forward(dw, low, fake, sv)                                   # :53
    gen_disc_loss = -score.mean()                                                      # :54
    gen_loss, reco_loss = gen_disc_loss, None                                          # :55-56
    if reconstruction_loss is not None:                                                # :57-59
        reco_loss = reconstruction_loss(low[..., :2], fake)
        gen_loss = gen_loss + reco_loss
    g_grads = _grads(gen_loss.sum(), gw, gk)                                           # :60 (non-scalar target: summed)
    with torch.no_grad():
        for k, v in st.items():
            gw[k].copy_(v)
    g_opt.step(gw, g_grads)                                                            # :61
    with torch.no_grad():                                                              # :63-68
        real_s = discriminator_forward(dw, low, high, sv)
        fake = generator_forward(gw, low, draws.noise(), False)
        fake_s = discriminator_forward(dw, low, fake, sv)
        d_loss = weighted_loss(loss_fn(real_s, fake_s), sample_weight)                 # :67
    return {
        "g_loss": -fake_s.mean(), "g_disc_loss": gen_disc_loss.detach(), "d_loss": d_loss,
        "g_reco_loss": None if reco_loss is None else reco_loss.detach(),
        "d_gradient_pen": gnorm.mean().detach(),
        "g_gradient_param": torch.stack([(g ** 2).mean() for g in g_grads.values()]).mean(),
        "d_gradient_param": torch.stack([(g ** 2).mean() for g in d_grads.values()]).mean(),
        "_d_loss_train": disc_loss, "fake": fake,
    }


def test_step(gw, dw, low, high, draws, d_loss_fn=None, shortcut_variant=False):
    """GAN.test_step, ganbase.py:96-113 (inference mode: no SN update, BatchNorm on moving statistics; the compiled loss is
    called WITHOUT the sample weights, :103).  Draw order: the generator noise first (:99)."""
    loss_fn = (lambda r, f: -(r.mean() - f.mean())) if d_loss_fn is None else d_loss_fn
    with torch.no_grad():
        noise = draws.noise()                                                          # :99
        true_predictions = discriminator_forward(dw, low, high, shortcut_variant)      # :100
        generated = generator_forward(gw, low, noise, False)                           # :101
        fake_predictions = discriminator_forward(dw, low, generated, shortcut_variant)  # :102
        return {"loss": loss_fn(true_predictions, fake_predictions), "generated": generated}   # :103


test_step.__test__ = False      # (not a pytest test)
