"""TEST INFRASTRUCTURE — numpy float64 restatement of the layer arithmetic, written from the layer
*definitions* (loops / index formulas), independent of torch.  It is the second opinion that pins
oracle/torch_model.py and oracle/torch_backend.py (which lean on torch functional ops with layout
conversions): tests/test_oracle_known_answers.py checks hand-computable known answers on these
functions and their agreement with the torch restatement.

Semantics restated (TensorFlow 2.4.3 / tensorflow-addons 0.14.0, third-party dependencies pinned in
/root/reference/requirements.txt:2-3, not vendored, not installable here) for the call sites of
/root/reference/src/downscaling/gan/models.py:28-140 — see SURVEY.md §8 a8.

PARITY UNPINNED against TensorFlow itself (no reference tests / golden vectors / runnable TF).
Only tests/ may import this module.
"""
import numpy as np


def leaky_relu(x, slope=0.2):
    return np.where(x > 0, x, slope * x)                       # max(x,0) + slope*min(x,0)


def hard_sigmoid(x):
    return np.clip(0.2 * x + 0.5, 0.0, 1.0)                    # Keras backend.hard_sigmoid


def zero_pad(x, p):
    return np.pad(x, ((0, 0), (p, p), (p, p), (0, 0)))         # ZeroPadding2D (symmetric)


def conv2d(x, w, b=None, stride=1, pad=0):
    """NHWC cross-correlation, HWIO kernel: y[n,i,j,o] = sum_{p,q,c} x[n, i*s+p-pad, j*s+q-pad, c] w[p,q,c,o]."""
    x = zero_pad(x, pad)
    n, H, W, C = x.shape
    kh, kw, _, O = w.shape
    Ho, Wo = (H - kh) // stride + 1, (W - kw) // stride + 1
    y = np.zeros((n, Ho, Wo, O))
    for p in range(kh):
        for q in range(kw):
            patch = x[:, p:p + (Ho - 1) * stride + 1:stride, q:q + (Wo - 1) * stride + 1:stride, :]
            y += patch @ w[p, q]
    return y if b is None else y + b


def conv2d_transpose(x, w, b=None, stride=1, crop=0):
    """Keras Conv2DTranspose with kernel (kh,kw,out,in): scatter form
    y[n, i*s+p, j*s+q, o] += x[n,i,j,c] * w[p,q,o,c], then `crop` pixels removed on every side
    ('valid': crop 0; 'same' with stride 1: crop (k-1)/2)."""
    n, H, W, C = x.shape
    kh, kw, O, _ = w.shape
    full = np.zeros((n, (H - 1) * stride + kh, (W - 1) * stride + kw, O))
    for p in range(kh):
        for q in range(kw):
            full[:, p:p + (H - 1) * stride + 1:stride, q:q + (W - 1) * stride + 1:stride, :] += x @ w[p, q].T
    if crop:
        full = full[:, crop:-crop, crop:-crop, :]
    return full if b is None else full + b


def upsample_bilinear_2x(x):
    """tf.image.resize(bilinear, half_pixel_centers): src = (dst+0.5)/2-0.5, lower=max(floor,0),
    upper=min(ceil,n-1), lerp=src-floor(src)."""
    def axis_weights(n):
        dst = np.arange(2 * n)
        src = (dst + 0.5) / 2.0 - 0.5
        fl = np.floor(src)
        lo = np.maximum(fl, 0).astype(int)
        hi = np.minimum(np.ceil(src), n - 1).astype(int)
        return lo, hi, src - fl
    n, H, W, C = x.shape
    lo, hi, t = axis_weights(H)
    x = x[:, lo] * (1 - t)[None, :, None, None] + x[:, hi] * t[None, :, None, None]
    lo, hi, t = axis_weights(W)
    return x[:, :, lo] * (1 - t)[None, None, :, None] + x[:, :, hi] * t[None, None, :, None]


def batch_norm_train(x, gamma, beta, eps=1e-3):
    ax = tuple(range(x.ndim - 1))
    mean, var = x.mean(ax), x.var(ax)                          # biased variance
    return (x - mean) / np.sqrt(var + eps) * gamma + beta, mean, var


def batch_norm_infer(x, gamma, beta, mean, var, eps=1e-3):
    return (x - mean) / np.sqrt(var + eps) * gamma + beta


def layer_norm(x, gamma, beta, eps=1e-3):
    mean = x.mean(-1, keepdims=True)
    var = x.var(-1, keepdims=True)
    return (x - mean) / np.sqrt(var + eps) * gamma + beta


def conv_lstm(x, kernel, rec, bias):
    """x [B,T,H,W,C] -> h_t for all t.  Gate order i,f,c,o on the last kernel axis."""
    B, T, H, W, _ = x.shape
    F = rec.shape[2]
    h = np.zeros((B, H, W, F))
    c = np.zeros((B, H, W, F))
    out = []
    for t in range(T):
        z = conv2d(x[:, t], kernel, bias, 1, 1) + conv2d(h, rec, None, 1, 1)
        i, f, g, o = z[..., :F], z[..., F:2 * F], z[..., 2 * F:3 * F], z[..., 3 * F:]
        c = hard_sigmoid(f) * c + hard_sigmoid(i) * np.tanh(g)
        h = hard_sigmoid(o) * np.tanh(c)
        out.append(h)
    return np.stack(out, 1)


def spectral_normalize(w, u):
    W = w.reshape(-1, w.shape[-1])

    def l2n(v):
        return v / np.sqrt(max((v * v).sum(), 1e-12))
    v = l2n(u @ W.T)
    un = l2n(v @ W)
    sigma = float((v @ W @ un.T).reshape(()))
    return w / sigma, un, sigma


def adam_tf(p, g, m, v, t, lr, b1, b2, eps):
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m = m + (1 - b1) * (g - m)
    v = v + (1 - b2) * (g * g - v)
    return p - lr_t * m / (np.sqrt(v) + eps), m, v


def wind_speed_weighted_rmse(real, fake):
    """gan/metrics.py:32-45."""
    u, v, uh, vh = real[..., 0], real[..., 1], fake[..., 0], fake[..., 1]
    est, rea = np.sqrt(uh ** 2 + vh ** 2), np.sqrt(u ** 2 + v ** 2)
    beta = (4 + rea) / (4 + est)
    tau = np.where(est >= rea, 0.425, 1 - 0.425)
    r = tau * ((uh - beta * u) ** 2 + (vh - beta * v) ** 2)
    r = np.where(np.isnan(r), 0.0, r)
    return np.sqrt(r.mean(axis=(1, 2, 3)))


def generator_forward(w, image, noise, training=False):
    """make_generator (gan/models.py:28-72) on numpy; w: {TF key: array}. BN in batch-statistics mode
    when training else moving statistics; SN is not applied here (weights taken as given)."""
    L = "layer_with_weights-"

    def td(fn, x):
        B, T = x.shape[:2]
        y = fn(x.reshape(B * T, *x.shape[2:]))
        return y.reshape(B, T, *y.shape[1:])

    def bn(x, i):
        k = L + str(i)
        if training:
            return batch_norm_train(x, w[k + "/gamma"], w[k + "/beta"])[0]
        return batch_norm_infer(x, w[k + "/gamma"], w[k + "/beta"], w[k + "/moving_mean"], w[k + "/moving_variance"])
    x = np.concatenate([image, noise], -1)
    x = bn(td(lambda a: leaky_relu(conv2d(a, w[L + "0/layer/w"], w[L + "0/layer/layer/bias"], 2, 3)), x), 1)
    res_2 = x
    x = bn(td(lambda a: leaky_relu(conv2d(a, w[L + "2/layer/w"], w[L + "2/layer/layer/bias"], 2, 1)), x), 3)
    res_4 = x
    x = conv_lstm(x, w[L + "4/cell/kernel"], w[L + "4/cell/recurrent_kernel"], w[L + "4/cell/bias"])
    x = bn(td(lambda a: leaky_relu(conv2d(a, w[L + "5/layer/w"], w[L + "5/layer/layer/bias"], 1, 1)), x), 6)
    x = np.concatenate([x, res_4], -1)
    x = bn(td(lambda a: leaky_relu(conv2d_transpose(a, w[L + "7/layer/w"], w[L + "7/layer/layer/bias"], 2, 0)), x), 8)
    x = np.concatenate([x, res_2], -1)
    x = td(lambda a: leaky_relu(conv2d_transpose(upsample_bilinear_2x(a), w[L + "9/layer/kernel"], w[L + "9/layer/bias"], 1, 2)), x)
    x = bn(x, 10)
    return td(lambda a: conv2d(a, w[L + "11/layer/kernel"], w[L + "11/layer/bias"], 1, 1), x)
