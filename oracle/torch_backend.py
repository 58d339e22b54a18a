"""TEST INFRASTRUCTURE — CPU oracle of the operator interface (never imported by the product).

`TorchOps` mirrors, method for method, `downscaling.engine.hipops.HipOps` (the HIP backend behind
the C ABI of include/wdgan.h) using stock torch-CPU functional ops, by default in float64.  Each
method restates the TF/Keras semantics of the reference layer it stands for (citations are to
/root/reference/src/downscaling; the arithmetic itself lives in tensorflow==2.4.3 /
tensorflow-addons==0.14.0, which are not vendored in the reference and not installed here):

  Conv2D / Conv2DTranspose / ConvLSTM2D convolutions ....... gan/models.py:33,39,45,49,55,64,70,93-134
  BatchNormalization (eps 1e-3, momentum .99, biased var) ... gan/models.py:34,40,50,56,69
  LayerNormalization (axis -1, eps 1e-3) ..................... gan/models.py:97,105,116,125,136
  LeakyReLU(0.2) as the conv activation ...................... gan/models.py:33 ff.
  UpSampling2D(2,'bilinear') = half-pixel, edge clamp ........ gan/models.py:62
  tfa SpectralNormalization, one power iteration ............. gan/models.py:33 ff.
  Adam in TF form (epsilon outside the bias correction) ...... gan/train.py:34-35,57-58

PARITY UNPINNED: the reference ships no tests, golden vectors or runnable TF, so nothing pins this
oracle to TensorFlow output; it is pinned only by hand-computed known-answer tests
(tests/test_oracle_known_answers.py) and by agreement with the independent numpy restatement
(oracle/np_ops.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from dataclasses import dataclass

import numpy as np
import math

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class ConvGeom:
    kh: int
    kw: int
    stride: int
    pad: int


class PackedWeights:
    def __init__(self, ops, w):
        kh, kw, cin, cout = w.shape
        self.ops, self.w = ops, w
        self.taps, self.cin, self.cout = kh * kw, cin, cout

    def refresh(self):
        pass

    def column_slice(self, n0, n1):
        return PackedWeights(self.ops, self.w[..., n0:n1])

    def as_1x1(self):
        return PackedWeights(self.ops, self.w.view(1, 1, self.taps * self.cin, self.cout))


def _lrelu(x, slope):
    return torch.where(x > 0, x, x * slope)


# ---- Philox4x32-10 (Salmon et al. 2011), vectorised over counters -------------------------------
def philox4x32_10(counter_lo, counter_hi, seed):
    """counter: uint64 arrays (lo 64 bits used as ctr[0:2]); returns uint32 array [..., 4]."""
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    c0 = (counter_lo & mask).astype(np.uint64)
    c1 = (counter_lo >> np.uint64(32)).astype(np.uint64)
    c2 = np.zeros_like(c0)
    c3 = np.zeros_like(c0)
    k0 = np.uint64(seed & 0xFFFFFFFF)
    k1 = np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & mask
        n1 = p1 & mask
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & mask
        n3 = p0 & mask
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def philox_normal_np(n, seed, offset):
    """n standard normals exactly as wdg_philox_normal orders them (Box-Muller on 24-bit uniforms)."""
    groups = (n + 3) // 4
    ctr = np.arange(groups, dtype=np.uint64) + np.uint64(offset)
    r = philox4x32_10(ctr, None, seed).astype(np.uint64)
    u = ((r >> np.uint64(8)) + np.uint64(1)).astype(np.float64) / 16777216.0  # (0, 1]
    z = np.empty((groups, 4), dtype=np.float64)
    for h in range(2):
        rad = np.sqrt(-2.0 * np.log(u[:, 2 * h]))
        ang = 2.0 * np.pi * u[:, 2 * h + 1]
        z[:, 2 * h] = rad * np.cos(ang)
        z[:, 2 * h + 1] = rad * np.sin(ang)
    return z.reshape(-1)[:n]


def philox_uniform_np(n, seed, offset):
    groups = (n + 3) // 4
    ctr = np.arange(groups, dtype=np.uint64) + np.uint64(offset)
    r = philox4x32_10(ctr, None, seed).astype(np.uint64)
    return ((r >> np.uint64(8)).astype(np.float64) / 16777216.0).reshape(-1)[:n]


class TorchOps:
    name = "torch-oracle"

    def __init__(self, dtype=torch.float64):
        self.dtype = dtype
        self.device = torch.device("cpu")

    # ---- plumbing ---------------------------------------------------------------------------
    def empty(self, *shape):
        return torch.zeros(*shape, dtype=self.dtype)

    def zeros(self, *shape, dtype=None):
        if dtype in (torch.int64, torch.int32):
            return torch.zeros(*shape, dtype=dtype)
        return torch.zeros(*shape, dtype=self.dtype)  # stats buffers are "fp64" on HIP; oracle dtype here

    def from_host(self, arr):
        return torch.as_tensor(np.asarray(arr)).to(self.dtype)

    def pack_weights(self, w):
        return PackedWeights(self, w)

    # ---- convolution family -----------------------------------------------------------------
    @staticmethod
    def _oihw(pk):
        return pk.w.permute(3, 2, 0, 1)

    @staticmethod
    def _bn_hook(out, bn_stats, bn_affine):
        """BatchNormalization hooks of the conv launches (wdg_conv_fwd_bn): statistics of the activated output, or the
        inference-mode scale / shift."""
        C = out.shape[-1]
        if bn_stats is not None:                      # slabs [2][round4(C)]
            Cp = bn_stats.shape[1] // 2
            bn_stats[0, :C] += out.sum((0, 1, 2)).to(bn_stats.dtype)
            bn_stats[0, Cp:Cp + C] += (out * out).sum((0, 1, 2)).to(bn_stats.dtype)
        if bn_affine is not None:                     # [scale | shift], round4(C) entries each
            Cp = bn_affine.shape[0] // 2
            out = out * bn_affine[:C] + bn_affine[Cp:Cp + C]
        return out

    def conv_fwd(self, x, pk, bias, y, g, act=False, accumulate=False, slope=0.2, bn_stats=None, bn_affine=None):
        xin = x[..., :pk.cin].permute(0, 3, 1, 2)
        out = F.conv2d(xin, self._oihw(pk), bias, stride=g.stride, padding=g.pad).permute(0, 2, 3, 1)
        if act:
            out = _lrelu(out, slope)
        out = self._bn_hook(out, bn_stats, bn_affine)
        if accumulate:
            out = out + y[..., :pk.cout]
        y[..., :pk.cout] = out

    def conv_fwd_ln(self, x, pk, bias, y, z, g, gamma, beta, eps, mean_rstd, act=True, slope=0.2):
        """conv -> bias -> LeakyReLU -> LayerNormalization (wdg_conv_fwd_ln): the two reference layers, one after the other."""
        self.conv_fwd(x, pk, bias, y, g, act=act, slope=slope)
        self.ln_fwd(y.view(-1, y.shape[-1])[:, :pk.cout], gamma, beta, eps, z.view(-1, z.shape[-1])[:, :pk.cout], mean_rstd)

    def conv_dgrad(self, dy, pk, dx, g, bias=None, act=False, accumulate=False, slope=0.2, bn_stats=None, bn_affine=None):
        H, W = dx.shape[1], dx.shape[2]
        Ho, Wo = dy.shape[1], dy.shape[2]
        oph = H - ((Ho - 1) * g.stride - 2 * g.pad + g.kh)
        opw = W - ((Wo - 1) * g.stride - 2 * g.pad + g.kw)
        out = F.conv_transpose2d(dy[..., :pk.cout].permute(0, 3, 1, 2), self._oihw(pk), bias, stride=g.stride,
                                 padding=g.pad, output_padding=(oph, opw)).permute(0, 2, 3, 1)
        if act:
            out = _lrelu(out, slope)
        out = self._bn_hook(out, bn_stats, bn_affine)
        if accumulate:
            out = out + dx[..., :pk.cin]
        dx[..., :pk.cin] = out

    def conv_dgrad_lnbwd(self, dy, pk, dx, g, y, mean_rstd, gamma, c0, C, act_slope, dgamma, dbeta, dbias, par_ws=None):
        """The operator interface's fused call (HipOps.conv_dgrad_lnbwd) as the composition it stands for: data gradient, then
        the LayerNorm + LeakyReLU backward on channels [c0, c0 + C) of it, in place."""
        self.conv_dgrad(dy, pk, dx, g)
        grp = dx[..., c0:c0 + C]
        d2 = grp.reshape(-1, C)
        out = torch.empty_like(d2)
        self.ln_bwd(d2, y[..., :C].reshape(-1, C), mean_rstd, gamma, act_slope, out, dgamma, dbeta, dbias)
        dx[..., c0:c0 + C] = out.reshape(grp.shape)

    def lnbwd_scratch(self, C):
        return None

    # inference precision: operands rounded to a 16-bit format (bf16 or IEEE fp16, round-to-nearest-even), exact accumulation
    @staticmethod
    def _r16(t, fmt="bf16"):
        return t.to(torch.bfloat16 if fmt == "bf16" else torch.float16).to(t.dtype)

    def conv_fwd_bf16(self, x, pk, bias, y, g, act=False, affine=None, accumulate=False, slope=0.2, fmt="bf16"):
        out = torch.zeros(y.shape[:3] + (pk.cout,), dtype=x.dtype)
        self.conv_fwd(self._r16(x, fmt), PackedWeights(self, self._r16(pk.w, fmt)), bias, out, g, act=act, slope=slope)
        if affine is not None:
            out = out * affine[:pk.cout] + affine[pk.cout:]
        y[..., :pk.cout] = out + (y[..., :pk.cout] if accumulate else 0)

    def conv_dgrad_bf16(self, dy, pk, dx, g, bias=None, act=False, affine=None, accumulate=False, slope=0.2, fmt="bf16"):
        out = torch.zeros(dx.shape[:3] + (pk.cin,), dtype=dy.dtype)
        self.conv_dgrad(self._r16(dy, fmt), PackedWeights(self, self._r16(pk.w, fmt)), out, g, bias=bias, act=act, slope=slope)
        if affine is not None:
            out = out * affine[:pk.cin] + affine[pk.cin:]
        dx[..., :pk.cin] = out + (dx[..., :pk.cin] if accumulate else 0)

    def conv_halo_fwd_bf16(self, x, pk, bias, y, g, act=False, affine=None, slope=0.2, fmt="bf16"):
        self.conv_fwd_bf16(x, pk, bias, y, g, act=act, affine=affine, slope=slope, fmt=fmt)

    upconv_colfwd = True    # mirrors HipOps.upconv_colfwd: which operand the 16-bit path rounds

    def upconv_fwd_bf16(self, x_low, pk, bias, y, g, act=True, affine=None, slope=0.2, fmt="bf16", pool=None):
        up = torch.zeros(x_low.shape[0], 2 * x_low.shape[1], 2 * x_low.shape[2], x_low.shape[3], dtype=x_low.dtype)
        if self.upconv_colfwd and (g.kh, g.kw, g.stride, g.pad) == (5, 5, 1, 2) and pk.cout % 8 == 0 and pk.cin in (4, 8, 16):
            # column form: the 16-bit GEMM operands are the LOW-RES input and the weights; the bilinear interpolation
            # acts on the exact products afterwards (it commutes with the channel mixing)
            if getattr(self, "z16", False):
                # the column GEMM's result z[r, (t, o)] = sum_c x[r, c] w[t][o][c] is itself stored in the 16-bit format
                # (wdg_upconv_colgemm_h16): per tap t, round z_t, interpolate it, shift it by the tap and add
                n, Hl, Wl, _ = x_low.shape
                xr, wr = self._r16(x_low[..., :pk.cout], fmt), self._r16(pk.w, fmt)       # w [5][5][o][c]
                acc = torch.zeros(n, 2 * Hl, 2 * Wl, pk.cin, dtype=x_low.dtype)
                for ta in range(5):
                    for tb in range(5):
                        zt = self._r16(torch.einsum("nhwc,oc->nhwo", xr, wr[ta, tb]), fmt)
                        ut = torch.zeros(n, 2 * Hl, 2 * Wl, pk.cin, dtype=x_low.dtype)
                        self.upsample2x_fwd(zt, ut)
                        # transposed conv: y[u + t - 2] += U(z_t)[u] (positions outside the image drop: 'same' padding)
                        dy, dx = ta - 2, tb - 2
                        ys, xs = slice(max(0, dy), 2 * Hl - max(0, -dy)), slice(max(0, dx), 2 * Wl - max(0, -dx))
                        yd, xd = slice(max(0, -dy), 2 * Hl - max(0, dy)), slice(max(0, -dx), 2 * Wl - max(0, dx))
                        acc[:, ys, xs] += ut[:, yd, xd]
                out = acc + (bias if bias is not None else 0)
                if act:
                    out = torch.where(out > 0, out, out * slope)
                if affine is not None:
                    out = out * affine[:pk.cin] + affine[pk.cin:]
                y[..., :pk.cin] = out
                return
            self.upsample2x_fwd(self._r16(x_low, fmt), up)
            out = torch.zeros(y.shape[:3] + (pk.cin,), dtype=x_low.dtype)
            self.conv_dgrad(up, PackedWeights(self, self._r16(pk.w, fmt)), out, g, bias=bias, act=act, slope=slope)
            if affine is not None:
                out = out * affine[:pk.cin] + affine[pk.cin:]
            y[..., :pk.cin] = out
            return
        # 25-tap halo kernel: bilinear interpolation in full precision, then 16-bit rounding of the conv operands
        self.upsample2x_fwd(x_low, up)
        self.conv_dgrad_bf16(up, pk, y, g, bias=bias, act=act, affine=affine, slope=slope, fmt=fmt)

    def upconv_fwd(self, x_low, pk, bias, y, g, act=True, slope=0.2, pool=None, bn_stats=None, bn_affine=None):
        up = torch.zeros(x_low.shape[0], 2 * x_low.shape[1], 2 * x_low.shape[2], x_low.shape[3], dtype=x_low.dtype)
        self.upsample2x_fwd(x_low, up)
        self.conv_dgrad(up, pk, y, g, bias=bias, act=act, slope=slope, bn_stats=bn_stats, bn_affine=bn_affine)

    def upconv_bwd(self, x_low, dpre, pk, dw, dx_low, g, pool=None):
        """Backward of upconv_fwd as the reference's tape computes it: through the materialised upsampled tensor."""
        n, Hl, Wl, C = x_low.shape
        up = torch.zeros(n, 2 * Hl, 2 * Wl, C, dtype=x_low.dtype)
        dup = torch.zeros(n, 2 * Hl, 2 * Wl, C, dtype=x_low.dtype)
        self.upsample2x_fwd(x_low, up)
        self.conv_wgrad(dpre, up, pk, dw, g, accumulate=True)
        self.conv_fwd(dpre, pk, None, dup, g, act=False)
        self.upsample2x_bwd(dup, dx_low)

    def conv_wgrad(self, x, dy, pk, dw, g, accumulate=True, dbias=None):
        xin = x[..., :pk.cin].permute(0, 3, 1, 2).contiguous()
        gout = dy[..., :pk.cout].permute(0, 3, 1, 2).contiguous()
        gw = torch.nn.grad.conv2d_weight(xin, (pk.cout, pk.cin, g.kh, g.kw), gout, stride=g.stride, padding=g.pad)
        gw = gw.permute(2, 3, 1, 0)
        if accumulate:
            dw += gw
        else:
            dw.copy_(gw)
        if dbias is not None:   # bias gradient of the same layer (always accumulated)
            dbias += dy[..., :pk.cout].reshape(-1, pk.cout).sum(0)

    def permute_bt(self, src, dst):
        C = min(src.shape[4], dst.shape[4])
        dst[..., :C] = src.permute(1, 0, 2, 3, 4)[..., :C]

    def fork(self, name="side", stream=None):
        """HipOps.fork restated for a backend without streams: run in place."""
        class _Inline:
            def __enter__(self):
                return self

            def __exit__(self, *exc):
                return False

            def join(self):
                pass
        return _Inline()

    # ---- evaluation metrics: restatement of /root/reference/src/downscaling/gan/metrics.py (the formulas, line by line) --
    def metrics_pointwise(self, real, fake):
        """[B, 6] per-sample sums (layout of wdg_metrics_pointwise): metrics.py:32-45, 79-88, 94-105, 66-73."""
        u, v, uh, vh = real[..., 0], real[..., 1], fake[..., 0], fake[..., 1]
        est, rea = torch.sqrt(uh ** 2 + vh ** 2), torch.sqrt(u ** 2 + v ** 2)
        beta = (4 + rea) / (4 + est)                                                   # :39-41 epsilon = 4, t = 0.425
        tau = torch.where(est >= rea, torch.full_like(u, 0.425), torch.full_like(u, 1 - 0.425))
        wsw = torch.nan_to_num(tau * ((uh - beta * u) ** 2 + (vh - beta * v) ** 2), nan=0.0, posinf=float("inf"), neginf=-float("inf"))
        wsr = torch.nan_to_num((rea - est) ** 2, nan=0.0, posinf=float("inf"), neginf=-float("inf"))
        nrm = lambda x: x * torch.rsqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=1e-12))   # noqa: E731 keras l2_normalize
        cos = (nrm(real) * nrm(fake)).sum(-1)
        acd = torch.acos(torch.clamp(cos, -1, 1)) / math.pi
        ocs = 0.5 * (1 - cos)
        sq = real ** 2
        ext = torch.nan_to_num(sq * (real - fake) ** 2, nan=0.0, posinf=float("inf"), neginf=-float("inf"))
        red = lambda x: x.double().flatten(1).sum(1)                                    # noqa: E731
        return torch.stack([red(wsw), red(wsr), red(acd), red(ocs), red(sq), red(ext)], 1)

    def lsd_sums(self, real, fake, eps):
        """metrics.py:121-137: tf.signal.rfft2d acts on the last two axes of the (B,T,H,W,C) tensor as written."""
        pr = torch.fft.rfft2(real).abs() ** 2 + eps
        pf = torch.fft.rfft2(fake).abs() ** 2 + eps
        ratio = torch.where(pf == 0, torch.zeros_like(pr), pr / pf)
        res = (10 * (torch.log(ratio) / math.log(10.0))) ** 2
        return res.double().flatten(1).sum(1), res[0].numel()

    def spatial_ks(self, real, fake, patch, points):
        """metrics.py:155-180: per (time, channel) image, stride-1 VALID patches, sup over `points` of |ECDF difference|."""
        pts = torch.as_tensor(points, dtype=real.dtype)
        stats = []
        for t in range(real.shape[1]):
            for c in range(real.shape[-1]):
                p1 = real[:, t, ..., c].unfold(1, patch, 1).unfold(2, patch, 1).flatten(-2)
                p2 = fake[:, t, ..., c].unfold(1, patch, 1).unfold(2, patch, 1).flatten(-2)
                ks = torch.zeros(p1.shape[:-1], dtype=torch.float64)
                for p in pts:
                    ks = torch.maximum(ks, ((p1 <= p).double().mean(-1) - (p2 <= p).double().mean(-1)).abs())
                stats.append(ks)
        return torch.stack(stats).mean(dim=(0, 1))

    def make_prep_batch(self, entries):
        """Restatement of HipOps.make_prep_batch: layer by layer (the batching is a launch-count optimisation)."""
        ops = self

        class _Batch:
            def run(self, sn, pack_all):
                for pk, u in entries:
                    if sn and u is not None:
                        ops.sn_power_iter(pk.w.view(-1, pk.cout), u.view(-1))
                    if pack_all or (sn and u is not None):
                        pk.refresh()
        return _Batch()

    def sn_power_iter(self, w2d, u):
        # tfa.layers.SpectralNormalization.normalize_weights, power_iterations=1
        def l2n(v):
            return v * torch.rsqrt(torch.clamp((v * v).sum(), min=1e-12))
        uu = u.reshape(1, -1)
        v = l2n(uu @ w2d.t())
        un = l2n(v @ w2d)
        sigma = (v @ w2d @ un.t()).reshape(())
        w2d.div_(sigma)
        u.copy_(un.reshape(-1))

    # ---- batch norm -------------------------------------------------------------------------
    def bn_stats(self, x, stats):
        C = x.shape[1]
        stats[:C] += x.sum(0).to(stats.dtype)
        stats[C:] += (x * x).sum(0).to(stats.dtype)

    def bn_finalize_train(self, stats, count, gamma, beta, mmean, mvar, momentum, eps, ss, saved):
        C = gamma.numel()
        stats = stats.sum(0)                                  # [R, 2C] replica slabs of the fused producers
        mean = stats[:C] / count
        var = torch.clamp(stats[C:] / count - mean * mean, min=0)
        inv = 1.0 / torch.sqrt(var + eps)
        ss[:C] = gamma * inv
        ss[C:] = beta - mean * gamma * inv
        saved[:C] = mean
        saved[C:] = inv
        mmean.mul_(momentum).add_(mean * (1 - momentum))
        # the fused op's batch_variance (what Keras averages into moving_variance) is Bessel-corrected in TF 2.4
        mvar.mul_(momentum).add_(var * (count / (count - 1.0) if count > 1 else 1.0) * (1 - momentum))

    def bn_collapse(self, stats):
        stats[0] = stats.sum(0)

    def bn_finalize_infer(self, gamma, beta, mmean, mvar, eps, ss):
        C = gamma.numel()
        inv = 1.0 / torch.sqrt(mvar + eps)
        ss[:C] = gamma * inv
        ss[C:] = beta - mmean * gamma * inv

    def bn_apply(self, x, ss, z):
        C = x.shape[1]
        z.copy_(x * ss[:C] + ss[C:])

    def bn_bwd_reduce(self, dz, y, saved, red):
        C = dz.shape[1]
        xh = (y - saved[:C]) * saved[C:]
        red[:C] += dz.sum(0)
        red[C:] += (dz * xh).sum(0)

    def bn_bwd_apply(self, dz, y, saved, gamma, red_mean, red_param, count, act_slope, dpre, dgamma, dbeta, dbias):
        C = dz.shape[1]
        xh = (y - saved[:C]) * saved[C:]
        d = gamma * saved[C:] * (dz - red_mean[:C] / count - xh * (red_mean[C:] / count))
        if act_slope >= 0:
            d = d * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, act_slope))
        dpre.copy_(d)
        if dbias is not None:
            dbias += d.sum(0)
        if red_param is not None:
            if dgamma is not None:
                dgamma += red_param[C:]
            if dbeta is not None:
                dbeta += red_param[:C]

    # ---- layer norm -------------------------------------------------------------------------
    def ln_fwd(self, y, gamma, beta, eps, z, mean_rstd):
        C = y.shape[1]
        z.copy_(F.layer_norm(y.contiguous(), (C,), gamma, beta, eps))
        if mean_rstd is not None:
            mean = y.mean(1)
            var = y.var(1, unbiased=False)
            mean_rstd[:, 0] = mean
            mean_rstd[:, 1] = 1.0 / torch.sqrt(var + eps)

    def ln_bwd(self, dz, y, mean_rstd, gamma, act_slope, dpre, dgamma, dbeta, dbias):
        xh = (y - mean_rstd[:, 0:1]) * mean_rstd[:, 1:2]
        g = dz * gamma
        d = mean_rstd[:, 1:2] * (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True))
        if act_slope >= 0:
            d = d * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, act_slope))
        if dgamma is not None:     # before the copy: dpre may alias dz
            dgamma += (dz * xh).sum(0)
        if dbeta is not None:
            dbeta += dz.sum(0)
        if dbias is not None:
            dbias += d.sum(0)
        dpre.copy_(d)

    # ---- ConvLSTM cell: Keras gate order i,f,c,o; hard_sigmoid = clip(.2x+.5,0,1) ------------
    @staticmethod
    def _hs(x):
        return torch.clamp(0.2 * x + 0.5, 0, 1)

    @staticmethod
    def _hs_grad(x):
        v = 0.2 * x + 0.5
        return torch.where((v >= 0) & (v <= 1), torch.full_like(x, 0.2), torch.zeros_like(x))

    def lstm_fwd(self, gates, c_prev, c, h, F_):
        gi, gf = self._hs(gates[:, :F_]), self._hs(gates[:, F_:2 * F_])
        gc, go = torch.tanh(gates[:, 2 * F_:3 * F_]), self._hs(gates[:, 3 * F_:4 * F_])
        cn = gi * gc
        if c_prev is not None:
            cn = cn + gf * c_prev[:, :F_]
        c[:, :F_] = cn
        h[:, :F_] = go * torch.tanh(cn)

    def lstm_bwd(self, gates, c_prev, c, dh, dc_in, dgates, dc_prev, F_):
        xi, xf, xc, xo = (gates[:, k * F_:(k + 1) * F_] for k in range(4))
        gi, gf, gc, go = self._hs(xi), self._hs(xf), torch.tanh(xc), self._hs(xo)
        cp = c_prev[:, :F_] if c_prev is not None else torch.zeros_like(gi)
        tc = torch.tanh(c[:, :F_])
        dhv = dh[:, :F_]
        dc = dhv * go * (1 - tc * tc)
        if dc_in is not None:
            dc = dc + dc_in[:, :F_]
        dgates[:, :F_] = dc * gc * self._hs_grad(xi)
        dgates[:, F_:2 * F_] = dc * cp * self._hs_grad(xf)
        dgates[:, 2 * F_:3 * F_] = dc * gi * (1 - gc * gc)
        dgates[:, 3 * F_:4 * F_] = dhv * tc * self._hs_grad(xo)
        if dc_prev is not None:
            dc_prev[:, :F_] = dc * gf

    def convlstm1_supported(self, cin, F_):
        return (cin, F_) in ((2, 2), (5, 16))

    def convlstm1_x2_supported(self, cin, F_, n2):
        return cin == 5 and F_ == 16 and 1 <= n2 <= 2

    @staticmethod
    def _with_x2(x, cin, x2):
        """The layer's input with its last n2 channels taken from the second tensor (HipOps.convlstm1_fwd / _bwd, x2=)."""
        if x2 is None:
            return x
        t2, n2 = x2
        x = x.clone()
        x[..., cin - n2:cin] = t2[..., :n2]
        return x

    def convlstm1_fwd(self, x, wx, bias, h, cin, F_, x2=None):
        x = self._with_x2(x, cin, x2)
        gates = torch.zeros(*x.shape[:3], 4 * F_, dtype=x.dtype)
        self.conv_fwd(x, PackedWeights(self, wx), bias, gates, ConvGeom(3, 3, 1, 1))
        c = torch.zeros(*x.shape[:3], F_, dtype=x.dtype)
        hh = torch.zeros(*x.shape[:3], F_, dtype=x.dtype)
        self.lstm_fwd(gates.view(-1, 4 * F_), None, c.view(-1, F_), hh.view(-1, F_), F_)
        h[..., :F_] = hh

    def convlstm1_dx_from_supported(self, cin, F_, c0):
        return cin == 5 and F_ == 16 and c0 == 3

    def convlstm1_bwd(self, x, wx, bias, dh, dgates, dx, cin, F_, accumulate_dx=False, dw=None, dbias=None, x2=None, dx_c0=0):
        if dx_c0:
            # HipOps.convlstm1_bwd(dx_c0=): the gradient of input channels [dx_c0, cin) only, into dx[..., :cin - dx_c0]
            full = torch.zeros(*x.shape[:3], (cin + 3) // 4 * 4, dtype=x.dtype)
            self.convlstm1_bwd(x, wx, bias, dh, None, full, cin, F_)
            part = full[..., dx_c0:cin]
            if accumulate_dx:
                dx[..., :cin - dx_c0] += part
            else:
                dx[..., :cin - dx_c0] = part
            return
        x = self._with_x2(x, cin, x2)
        pk = PackedWeights(self, wx)
        gates = torch.zeros(*x.shape[:3], 4 * F_, dtype=x.dtype)
        self.conv_fwd(x, pk, bias, gates, ConvGeom(3, 3, 1, 1))
        c = torch.zeros(*x.shape[:3], F_, dtype=x.dtype)
        hh = torch.zeros(*x.shape[:3], F_, dtype=x.dtype)
        self.lstm_fwd(gates.view(-1, 4 * F_), None, c.view(-1, F_), hh.view(-1, F_), F_)
        dg = torch.zeros_like(gates)
        self.lstm_bwd(gates.view(-1, 4 * F_), None, c.view(-1, F_), dh[..., :F_].reshape(-1, F_), None,
                      dg.view(-1, 4 * F_), None, F_)
        if dgates is not None:
            dgates.copy_(dg)
        if dw is not None:
            self.conv_wgrad(x, dg, pk, dw, ConvGeom(3, 3, 1, 1), accumulate=True, dbias=dbias)
        if dx is not None:
            self.conv_dgrad(dg, pk, dx, ConvGeom(3, 3, 1, 1), accumulate=accumulate_dx)

    def convln_supported(self, cin, cout):
        return cout == 16 and cin == 2

    def convln_fwd(self, x, w, bias, gamma, beta, eps, slope, y, z, mean_rstd):
        C = w.shape[3]
        if y is None:       # z only: the backward (convln_bwd_x) recomputes y and the statistics from x
            y, mean_rstd = torch.zeros(z.shape, dtype=z.dtype), torch.zeros(z.numel() // C, 2, dtype=z.dtype)
        self.conv_fwd(x, PackedWeights(self, w), bias, y, ConvGeom(3, 3, 1, 1), act=True, slope=slope)
        zz = torch.zeros(y.shape, dtype=y.dtype)
        self.ln_fwd(y.reshape(-1, C), gamma, beta, eps, zz.view(-1, C), mean_rstd)
        z.copy_(zz)

    def convln_bwd_x(self, dz, x, w, bias, gamma, eps, slope, dx, dgamma, dbeta, dbias, dw):
        C = w.shape[3]
        pk = PackedWeights(self, w)
        y = torch.zeros(dz.shape, dtype=dz.dtype)
        self.conv_fwd(x, pk, bias, y, ConvGeom(3, 3, 1, 1), act=True, slope=slope)
        mr = torch.zeros(y.numel() // C, 2, dtype=dz.dtype)
        self.ln_fwd(y.reshape(-1, C), gamma, torch.zeros_like(gamma), eps, torch.zeros(y.numel() // C, C, dtype=dz.dtype), mr)
        dpre = torch.zeros(dz.shape, dtype=dz.dtype)
        self.ln_bwd(dz.reshape(-1, C), y.reshape(-1, C), mr, gamma, slope, dpre.view(-1, C), dgamma, dbeta, dbias)
        if dw is not None:
            self.conv_wgrad(x, dpre, pk, dw, ConvGeom(3, 3, 1, 1), accumulate=True)
        if dx is not None:
            self.conv_dgrad(dpre, pk, dx, ConvGeom(3, 3, 1, 1))

    def convln_bwd(self, dz, y, mean_rstd, w, gamma, slope, dpre, dx, dgamma, dbeta, dbias):
        C = w.shape[3]
        self.ln_bwd(dz.reshape(-1, C), y.reshape(-1, C), mean_rstd, gamma, slope, dpre.view(-1, C), dgamma, dbeta, dbias)
        if dx is not None:
            self.conv_dgrad(dpre, PackedWeights(self, w), dx, ConvGeom(3, 3, 1, 1))

    # ---- resampling / head ------------------------------------------------------------------
    def upsample2x_fwd(self, x, y):
        out = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False)
        y.copy_(out.permute(0, 2, 3, 1))

    def upsample2x_bwd(self, dy, dx, accumulate=False):
        xin = torch.zeros(dx.shape, dtype=dy.dtype).permute(0, 3, 1, 2).requires_grad_(True)
        out = F.interpolate(xin, scale_factor=2, mode="bilinear", align_corners=False)
        (gx,) = torch.autograd.grad(out, xin, dy.permute(0, 3, 1, 2))
        gx = gx.permute(0, 2, 3, 1)
        if accumulate:
            dx += gx
        else:
            dx.copy_(gx)

    def patch_gather(self, x, out, k, stride, pad):
        n, H, W, Cc = x.shape
        t = out.shape[1]
        xp = F.pad(x, (0, 0, pad, max(0, (t - 1) * stride + k - pad - W), pad, max(0, (t - 1) * stride + k - pad - H)))
        for oy in range(t):
            for ox in range(t):
                out[:, oy, ox] = xp[:, oy * stride:oy * stride + k, ox * stride:ox * stride + k, :].reshape(n, -1)

    def patch_scatter(self, dpatch, dx, k, stride, pad, accumulate=False):
        n, H, W, Cc = dx.shape
        t = dpatch.shape[1]
        Hp, Wp = max(H + pad, (t - 1) * stride + k), max(W + pad, (t - 1) * stride + k)
        buf = torch.zeros(n, Hp, Wp, Cc, dtype=dx.dtype)
        for oy in range(t):
            for ox in range(t):
                buf[:, oy * stride:oy * stride + k, ox * stride:ox * stride + k, :] += dpatch[:, oy, ox].reshape(n, k, k, Cc)
        g = buf[:, pad:pad + H, pad:pad + W, :]
        if accumulate:
            dx += g
        else:
            dx.copy_(g)

    def dense_gap_fwd(self, x, w, b, score, B, T):
        s = (x @ w + b[0]).reshape(T, B)  # rows are time-major
        score.copy_(s.mean(0))

    def dense_gap_bwd_ln(self, x, w, dscore, dx, dw, db, B, T, y, mean_rstd, gamma, C, act_slope, dgamma, dbeta, dbias, par_ws=None):
        """HipOps.dense_gap_bwd_ln as the composition it stands for."""
        self.dense_gap_bwd(x, w, dscore, dx, dw, db, B, T)
        d2 = dx.reshape(-1, C)
        out = torch.empty_like(d2)
        self.ln_bwd(d2, y.reshape(-1, C), mean_rstd, gamma, act_slope, out, dgamma, dbeta, dbias)
        dx.copy_(out.reshape(dx.shape))

    def dense_gap_bwd(self, x, w, dscore, dx, dw, db, B, T):
        drow = (dscore.reshape(1, B) / T).expand(T, B).reshape(-1)
        if dx is not None:
            dx.copy_(drow[:, None] * w[None, :])
        if dw is not None:
            dw += x.t() @ drow
        if db is not None:
            db += dscore.sum()

    # ---- elementwise / reductions -----------------------------------------------------------
    def copy_channels(self, src, dst, accumulate=False):
        if accumulate:
            dst += src
        else:
            dst.copy_(src)

    def colsum(self, x, out, accumulate=True):
        if accumulate:
            out += x.sum(0)
        else:
            out.copy_(x.sum(0))

    def lerp_batch(self, a, b, eps, out, pixels_per_img, B):
        P = a.shape[0]
        bi = (torch.arange(P) // pixels_per_img) % B
        e = eps[bi][:, None]
        out.copy_(e * a + (1 - e) * b)

    def sumsq_batch_ch(self, x, pixels_per_img, T, B, out):
        Cc = x.shape[1]
        out.copy_((x * x).reshape(T, B, pixels_per_img, Cc).sum((0, 2)))

    def segment_meansq(self, flat, offsets, out):
        for i in range(out.numel()):
            seg = flat[int(offsets[2 * i]):int(offsets[2 * i + 1])]
            out[i] = (seg * seg).mean()

    def philox_normal(self, out, seed, offset, std, add=None):
        P, Cc = out.shape
        z = torch.from_numpy(philox_normal_np(P * Cc, seed, offset)).to(self.dtype).reshape(P, Cc)
        out.copy_(std * z if add is None else add + std * z)

    def philox_uniform(self, out, seed, offset):
        out.copy_(torch.from_numpy(philox_uniform_np(out.numel(), seed, offset)).to(self.dtype).reshape(out.shape))

    def zero_ranges(self, flat, ranges):
        for a, b in ranges:
            flat[a:b].zero_()

    def adam_tf(self, p, g, m, v, lr_t, beta1, beta2, eps, grad_scale=1.0):
        gg = g * grad_scale
        m += (1 - beta1) * (gg - m)
        v += (1 - beta2) * (gg * gg - v)
        p -= lr_t * m / (torch.sqrt(v) + eps)
