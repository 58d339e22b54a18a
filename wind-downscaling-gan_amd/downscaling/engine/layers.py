"""Layer objects with explicit forward / backward, written against the operator interface
(`HipOps` on the GPU; tests inject the float64 CPU oracle backend to check this host logic).

There is no autograd graph: the generator and discriminator are fixed graphs
(/root/reference/src/downscaling/gan/models.py:9-142), so each layer stores exactly what its own
backward needs and the networks call the backward passes in reverse order.  Layer order of
operations follows the reference: conv -> bias -> LeakyReLU(0.2) -> norm; the LeakyReLU derivative
is taken from the sign of the stored activation and fused into the norm backward.
"""
import os

from . import params as P
from .common import ConvGeom, round4, v2

LRELU = 0.2
BN_EPS = 1e-3      # keras.layers.BatchNormalization default
BN_MOMENTUM = 0.99
LN_EPS = 1e-3      # keras.layers.LayerNormalization default


class Conv:
    """Conv2D or Conv2DTranspose (+ optional tfa SpectralNormalization wrapper).

    For `transposed=True` (cin, cout) are those of the convolution the layer is the adjoint of, so
    the TF kernel (kh, kw, out, in) is stored unchanged as HWIO with I = layer outputs, O = layer
    inputs; forward runs the data-gradient kernel, input-gradient runs the forward kernel.
    """

    def __init__(self, net, name, k, cin, cout, stride, pad, *, transposed=False, sn=False, act=True, prefix="layer",
                 cout_gap=None):
        """cout_gap = (pos, width): the `cout` side of the kernel (for a transposed layer: the channels of the tensor it
        reads) carries `width` zero alignment channels at `pos`; `cout` counts the real channels (params.Var.gap)."""
        self.net, self.ops, self.name = net, net.ops, name
        self.g = ConvGeom(k, k, stride, pad)
        self.cin, self.cout, self.transposed, self.sn, self.act = cin, cout, transposed, sn, act
        store = net.params
        if cout_gap is not None:
            assert not sn and transposed, "the gap is on the reduction side of a plain transposed layer only"
            self.w = store.add(f"{name}/{prefix}/kernel", (k, k, cin, cout), P.conv_glorot, gap=(3, cout_gap[0], cout_gap[1]))
            self.b = store.add(f"{name}/{prefix}/bias", (cin,), P.zeros_init)
            self.u, self.pk = None, None
            self.cout = cout + cout_gap[1]
            return
        # TF checkpoint keys (weights-55.ckpt/*.index): TimeDistributed -> "layer", SN wrapper -> "w"/"sn_u"
        wname = f"{name}/{prefix}/w" if sn else f"{name}/{prefix}/kernel"
        bname = f"{name}/{prefix}/layer/bias" if sn else f"{name}/{prefix}/bias"
        self.w = store.add(wname, (k, k, cin, cout), P.conv_glorot)
        self.b = store.add(bname, (cin if transposed else cout,), P.zeros_init)
        self.u = store.add(f"{name}/{prefix}/sn_u", (1, cout), P.sn_u_init, trainable=False) if sn else None
        self.pk = None
        self.w.lazy_ok = not transposed          # (backward_weights below is this kernel gradient's only writer)

    def build(self):
        self.pk = self.ops.pack_weights(self.w.value)

    def prep_entries(self):
        """(packed weights, sn_u or None) for the network's batched SN / repack stage (_Net._prepare)."""
        return [(self.pk, self.u.value.view(-1) if self.sn else None)]

    def forward(self, x, y, bn_stats=None, bn_affine=None):
        """bn_stats / bn_affine: the BatchNormalization that follows the layer, folded into the launch (training: batch
        statistics of y accumulated by the epilogue; inference: y normalised in the epilogue) — HipOps.conv_fwd."""
        hooks = {} if bn_stats is None and bn_affine is None else dict(bn_stats=bn_stats, bn_affine=bn_affine)
        if self.transposed:
            self.ops.conv_dgrad(x, self.pk, y, self.g, bias=self.b.value, act=self.act, slope=LRELU, **hooks)
        else:
            self.ops.conv_fwd(x, self.pk, self.b.value, y, self.g, act=self.act, slope=LRELU, **hooks)

    def forward_ln(self, x, y, z, ln):
        """conv -> bias -> LeakyReLU -> LayerNormalization `ln` in one operator call (the norm rides in the conv's epilogue
        where one owns complete rows): y keeps the pre-norm activation for the backward pass, z the normalised output."""
        assert not self.transposed and self.act
        ln.ensure_stats(y.shape[0] * y.shape[1] * y.shape[2])
        self.ops.conv_fwd_ln(x, self.pk, self.b.value, y, z, self.g, ln.gamma.value, ln.beta.value, LN_EPS, ln.mean_rstd,
                             act=True, slope=LRELU)

    def forward_bf16(self, x, y, affine=None, fmt="bf16"):
        """Inference precision (bf16 or fp16 operands, fp32 accumulate); `affine` = fused inference BatchNorm."""
        if self.transposed:
            self.ops.conv_dgrad_bf16(x, self.pk, y, self.g, bias=self.b.value, act=self.act, affine=affine, slope=LRELU, fmt=fmt)
        else:
            self.ops.conv_fwd_bf16(x, self.pk, self.b.value, y, self.g, act=self.act, affine=affine, slope=LRELU, fmt=fmt)

    def backward_weights(self, x, dpre):
        # (a gradient slot left unfilled by ParamStore.zero_grad(lazy=True) is stored to, not accumulated into)
        acc = not self.w.fresh
        self.w.fresh = False
        if self.transposed:
            self.ops.conv_wgrad(dpre, x, self.pk, self.w.grad, self.g, accumulate=acc)
        else:
            self.ops.conv_wgrad(x, dpre, self.pk, self.w.grad, self.g, accumulate=acc)

    def backward_input(self, dpre, dx, accumulate=False):
        if self.transposed:
            self.ops.conv_fwd(dpre, self.pk, None, dx, self.g, act=False, accumulate=accumulate)
        else:
            self.ops.conv_dgrad(dpre, self.pk, dx, self.g, accumulate=accumulate)

    def backward_input_through_ln(self, dpre, dx, ln, y, c0, dbias, need_param_grads, act_slope=LRELU):
        """backward_input, chained with the backward of the LayerNormalization `ln` (+ LeakyReLU) that produced channels
        [c0, c0 + ln.C) of this layer's input from the pre-norm activation `y`: those channels of dx leave the call as the
        gradient w.r.t. the producer's pre-activation (what ln.backward would make of them in a pass of its own), and the
        norm's / the producer's bias gradients are accumulated — HipOps.conv_dgrad_lnbwd."""
        assert not self.transposed
        if need_param_grads:
            self.ops.conv_dgrad_lnbwd(dpre, self.pk, dx, self.g, y, ln.mean_rstd, ln.gamma.value, c0, ln.C, act_slope,
                                      ln.gamma.grad, ln.beta.grad, dbias, ln.lnbwd_scratch())
        else:
            self.ops.conv_dgrad_lnbwd(dpre, self.pk, dx, self.g, y, ln.mean_rstd, ln.gamma.value, c0, ln.C, act_slope,
                                      None, None, None, None)


class Dense:
    """keras.layers.Dense (linear) under TimeDistributed: rows = images, run as a 1x1 convolution on an [N,1,1,K] view
    (kernel [K, units] viewed as HWIO [1,1,K,units])."""
    G1 = ConvGeom(1, 1, 1, 0)

    def __init__(self, net, name, K, units, prefix="layer"):
        self.net, self.ops, self.K, self.units = net, net.ops, K, units
        self.w = net.params.add(f"{name}/{prefix}/kernel", (K, units), P.glorot_uniform(K, units))
        self.b = net.params.add(f"{name}/{prefix}/bias", (units,), P.zeros_init)
        self.pk = None

    def build(self):
        self.pk = self.ops.pack_weights(self.w.value.view(1, 1, self.K, self.units))

    def prep_entries(self):
        return [(self.pk, None)]

    def forward(self, x, y):
        """x [N,1,1,>=K], y [N,1,1,round4(units)]."""
        self.ops.conv_fwd(x, self.pk, self.b.value, y, self.G1, act=False)

    def backward_input(self, dy, dx):
        self.ops.conv_dgrad(dy, self.pk, dx, self.G1)


class BatchNorm:
    """keras BatchNormalization over the channel axis.  A channel count that is not a multiple of 4 (feature_channels / 8
    for feature_channels % 32 != 0) runs at the padded width: activations carry zero pad channels, and the per-channel
    vectors are read through the 4-element alignment slots of the flat parameter buffers (zeros beyond C: scale 0,
    shift 0, so the pad channels stay zero and their gradient slots receive zeros)."""
    REPLICAS = 512

    def __init__(self, net, name, C):
        self.net, self.ops, self.C, self.Cp = net, net.ops, C, round4(C)
        st = net.params
        self.gamma = st.add(f"{name}/gamma", (C,), P.ones_init)
        self.beta = st.add(f"{name}/beta", (C,), P.zeros_init)
        self.mmean = st.add(f"{name}/moving_mean", (C,), P.zeros_init, trainable=False)
        self.mvar = st.add(f"{name}/moving_variance", (C,), P.ones_init, trainable=False)

    def build(self):
        import torch
        o, C = self.ops, self.Cp
        # [replica][sum | sum of squares]: the conv epilogues that produce this layer's input spread their fp64 atomics over
        # the replicas; wdg_bn_stats (standalone pass) uses replica 0; bn_finalize_train sums them
        self.stats = o.zeros(self.REPLICAS, 2 * C, dtype=torch.float64)
        self.red = o.zeros(2 * C, dtype=torch.float64)
        self.red_local = o.zeros(2 * C, dtype=torch.float64)
        self.ss = o.empty(2 * C)                 # training-mode [scale | shift] of the current pass
        self.ss_infer = o.empty(2 * C)           # inference-mode affine (moving statistics): its own buffer, so a training
                                                 # pass cannot clobber what a captured inference graph reads
        self.saved = o.empty(2 * C)
        self.count = 1.0
        self._epoch = 0                 # training-mode passes seen (each rewrites ss and the moving statistics)
        self._infer_key = None          # (params.version, _epoch) self.ss_infer holds the inference affine of
        if self.Cp != self.C:
            self.mvar.value_pad[self.C:] = 0.0     # (alignment slots: keep the pad channels' variance at 0, not the init 1)

    def begin_stats(self):
        """Zeroed statistics slabs for the producing conv launch (pass them as its bn_stats; then forward(..., have_stats=True))."""
        self.stats.zero_()
        return self.stats

    def forward(self, y, z, training, have_stats=False):
        o = self.ops
        if training:
            if not have_stats:
                self.stats.zero_()
                o.bn_stats(y, self.stats[0])
            count = float(y.shape[0])
            sync = self.net.sync
            stats = self.stats
            if sync is not None:  # SyncBN: the batch statistics couple the *global* batch
                o.bn_collapse(stats)          # the exchange moves 2*C values, not the replica slabs
                stats = stats[:1]
                sync.all_reduce_sum(stats)
                count *= sync.world_size
            self.count = count
            self._epoch += 1
            o.bn_finalize_train(stats, count, self.gamma.value_pad, self.beta.value_pad, self.mmean.value_pad,
                                self.mvar.value_pad, BN_MOMENTUM, BN_EPS, self.ss, self.saved)
            o.bn_apply(y, self.ss, z)
        else:
            o.bn_apply(y, self.infer_affine(), z)

    def infer_affine(self):
        """[scale | shift] of the inference-mode normalisation (moving statistics), for epilogue fusion.  Recomputed only when
        the parameters (ParamStore.version: optimizer step, load, set_weights) or the moving statistics (a training-mode pass)
        changed: five launches per generator forward otherwise — inside the replayed inference graph as well.  Anything that
        writes gamma / beta / the moving statistics in place by another road must call ParamStore.touch()."""
        key = (self.net.params.version, self._epoch)
        if self._infer_key != key:
            self.ops.bn_finalize_infer(self.gamma.value_pad, self.beta.value_pad, self.mmean.value_pad, self.mvar.value_pad, BN_EPS,
                                       self.ss_infer)
            self._infer_key = key
        return self.ss_infer

    def backward(self, dz, y, dpre, dbias, act_slope=LRELU):
        """dpre = BN-backward(dz) * lrelu'(y); accumulates dgamma, dbeta and (fused) the conv bias grad."""
        o = self.ops
        self.red.zero_()
        o.bn_bwd_reduce(dz, y, self.saved, self.red)
        red_param = self.red
        sync = self.net.sync
        if sync is not None:
            self.red_local.copy_(self.red)
            red_param = self.red_local
            sync.all_reduce_sum(self.red)
        o.bn_bwd_apply(dz, y, self.saved, self.gamma.value_pad, self.red, red_param, self.count, act_slope, dpre,
                       self.gamma.grad_pad, self.beta.grad_pad, dbias)


class LayerNorm:
    def __init__(self, net, name, C):
        self.net, self.ops, self.C = net, net.ops, C
        st = net.params
        self.gamma = st.add(f"{name}/gamma", (C,), P.ones_init)
        self.beta = st.add(f"{name}/beta", (C,), P.zeros_init)
        self.mean_rstd = None

    def ensure_stats(self, P):
        if self.mean_rstd is None or self.mean_rstd.shape[0] != P:
            self.mean_rstd = self.ops.empty(P, 2)

    def lnbwd_scratch(self):
        """Zeroed replica slabs for the parameter gradients of the fused data-gradient + LayerNorm-backward launch (owned by the
        layer: the twin discriminator runs its passes on another stream at the same time)."""
        if getattr(self, "_lnb_ws", None) is None:
            self._lnb_ws = self.ops.lnbwd_scratch(self.C)
        return self._lnb_ws

    def forward(self, y, z, keep_stats=True):
        self.ensure_stats(y.shape[0])
        self.ops.ln_fwd(y, self.gamma.value, self.beta.value, LN_EPS, z, self.mean_rstd)

    def backward(self, dz, y, dpre, dbias, need_param_grads, act_slope=LRELU):
        if need_param_grads:
            self.ops.ln_bwd(dz, y, self.mean_rstd, self.gamma.value, act_slope, dpre, self.gamma.grad,
                            self.beta.grad, dbias)
        else:
            self.ops.ln_bwd(dz, y, self.mean_rstd, self.gamma.value, act_slope, dpre, None, None, None)


class ConvLSTM:
    """keras.layers.ConvLSTM2D(F, 3x3, 'same', return_sequences=True) on time-major activations.

    gates = conv(x_t, kernel) + bias + conv(h_{t-1}, recurrent_kernel); order i,f,c,o; recurrent
    activation hard_sigmoid, activation tanh; h_0 = c_0 = 0, so the recurrent convolution and the
    forget path are skipped at t = 0.  The input convolution runs for all T timesteps in one launch.
    """

    def __init__(self, net, name, cin, F):
        self.net, self.ops, self.cin, self.F = net, net.ops, cin, F
        st = net.params
        self.wx = st.add(f"{name}/cell/kernel", (3, 3, cin, 4 * F), P.conv_glorot)
        self.wh = st.add(f"{name}/cell/recurrent_kernel", (3, 3, F, 4 * F), P.orthogonal)
        self.b = st.add(f"{name}/cell/bias", (4 * F,), P.lstm_bias(F))
        self.g = ConvGeom(3, 3, 1, 1)

    def build(self):
        self.pkx = self.ops.pack_weights(self.wx.value)
        self.pkh = self.ops.pack_weights(self.wh.value)
        self._shape = None
        self._pk_i = None
        self._chain_graphs = {}      # captured time loops of THIS layer (HipOps.chain)

    def prep_entries(self):
        return [(self.pkx, None), (self.pkh, None)]

    def _buffers(self, N, H, W):
        if self._shape != (N, H, W):
            o, F = self.ops, self.F
            self.gates = o.empty(N, H, W, 4 * F)
            self.c = o.empty(N, H, W, F)
            self.dgates = None
            self.dgates1 = None
            self._pk_i = None
            self._chain_graphs.clear()   # (captured on the previous buffers)
            self._shape = (N, H, W)

    def _fused1(self, T):
        """Single timestep + few channels: the fused, gate-recomputing kernels (convlstm1.hip)."""
        return T == 1 and self.ops.convlstm1_supported(self.cin, self.F)

    def x2_ok(self, T, n2):
        """Can this layer read its last n2 input channels from a second tensor (forward(..., x2=) / backward(..., x2=))?"""
        return self._fused1(T) and hasattr(self.ops, "convlstm1_x2_supported") and self.ops.convlstm1_x2_supported(self.cin, self.F, n2)

    def forward(self, x, h, B, T, bf16=False, fmt="bf16", x2=None):
        """x: [T*B,H,W,>=cin] view; h: [T*B,H,W,round4(F)] output buffer (pad channels stay zero).
        bf16=True: the two convolutions run at inference precision (the cell math stays fp32).
        x2 = (tensor, n2) (x2_ok layers only): the last n2 input channels are read from that tensor instead of x."""
        o, F = self.ops, self.F
        N, H, W, _ = h.shape
        if self._fused1(T):
            if x2 is not None:
                o.convlstm1_fwd(x, self.wx.value, self.b.value, h, self.cin, F, x2=x2)
            else:
                o.convlstm1_fwd(x, self.wx.value, self.b.value, h, self.cin, F)
            return
        assert x2 is None
        self._buffers(N, H, W)
        if not bf16 and T > 1:
            self._gates_x(x, T)
            self._time_loop_fwd(h, B, T)
            return
        if bf16 and T > 1 and hasattr(o, "convlstm16_supported") and \
                o.convlstm16_supported(x[:B], self.gates[:B], self.pkh, self.g, F) and \
                o.convlstm16_supported(x, self.gates, self.pkx, self.g, F):
            # inference precision: the input part of the gates with interleaved gate columns, then ONE launch per timestep
            # (recurrent convolution with the cell update in its epilogue) — csrc/conv_patch_h16.hip
            o.convlstm16_gates(x, self.pkx, self.b.value, self.gates, self.g, F, fmt=fmt)
            for t in range(T):
                sl, pv = slice(t * B, (t + 1) * B), slice((t - 1) * B, t * B)
                o.convlstm16_step(h[pv] if t else None, self.pkh, self.gates[sl], self.c[pv] if t else None, self.c[sl], h[sl],
                                  self.g, F, fmt=fmt)
            return
        conv = (lambda *a, **k: o.conv_fwd_bf16(*a, fmt=fmt, **k)) if bf16 else o.conv_fwd
        if T == 1 and not bf16:
            # h_0 = c_0 = 0: the forget gate is never read at t = 0 -> skip its quarter of the input convolution
            # (the slab keeps zeros there, so the backward's dgates_f = dc * c_prev * hs' = 0 is consistent)
            if self._pk_i is None:
                self._pk_i, self._pk_co = self.pkx.column_slice(0, F), self.pkx.column_slice(2 * F, 4 * F)
                self.gates[..., F:2 * F].zero_()
            conv(x, self._pk_i, self.b.value[:F], self.gates[..., :F], self.g, act=False)
            conv(x, self._pk_co, self.b.value[2 * F:], self.gates[..., 2 * F:], self.g, act=False)
        elif not bf16 and hasattr(o, "convlstm_gates_x_supported") and o.convlstm_gates_x_supported(x, self.gates, self.cin, F):
            # (the 5 -> 16 layer: 45-row reduction on the matrix pipe with the weights in registers, csrc/convlstm1.hip)
            o.convlstm_gates_x(x, self.wx.value, self.b.value, self.gates, self.cin, F)
        else:
            conv(x, self.pkx, self.b.value, self.gates, self.g, act=False)
        # fp32, T > 1: the recurrent convolution with the cell update in its epilogue where the halo-tile kernel runs the layer
        # (the discriminator's 2- and 16-feature ConvLSTMs): one launch per timestep instead of two
        step1 = (not bf16) and T > 1 and hasattr(o, "convlstm_step_supported") and \
            o.convlstm_step_supported(h[:B], self.gates[:B], self.pkh, self.g, F)
        def time_loop():
            for t in range(T):
                sl = slice(t * B, (t + 1) * B)
                if t > 0:
                    pv = slice((t - 1) * B, t * B)
                    if step1:
                        o.convlstm_step(h[pv], self.pkh, self.gates[sl], self.c[pv], self.c[sl], h[sl], self.g, F)
                        continue
                    conv(h[pv], self.pkh, None, self.gates[sl], self.g, act=False, accumulate=True)
                    o.lstm_fwd(v2(self.gates[sl]), v2(self.c[pv]), v2(self.c[sl]), v2(h[sl]), F)
                else:
                    o.lstm_fwd(v2(self.gates[sl]), None, v2(self.c[sl]), v2(h[sl]), F)

        # the recurrence is a chain of T (or 2T) small dependent launches on this layer's resident buffers: replayed from a HIP
        # graph where the backend offers it (HipOps.chain)
        packs = ()
        if step1 and hasattr(o, "convlstm_step_prepare"):
            packs = o.convlstm_step_prepare(h[:B], self.pkh, self.gates[:B], self.g, F) or ()
        if T > 2 and not bf16 and hasattr(o, "chain"):
            # (graphs are owned by this layer: they hold its buffers' addresses and die with them)
            o.chain(("lstm_fwd", h.data_ptr(), tuple(h.shape), self.gates.data_ptr(), self.c.data_ptr(), self.pkh.wF.data_ptr(),
                     B, T, bool(bf16), fmt, bool(step1)) + tuple(packs), time_loop, graphs=self._chain_graphs)
        else:
            time_loop()

    def dx_from_ok(self, T, c0, need_wgrad, x2=None):
        """Can backward(..., dx_c0=c0) produce the gradient of the input channels [c0, cin) alone?"""
        return (self._fused1(T) and not need_wgrad and x2 is None and hasattr(self.ops, "convlstm1_dx_from_supported")
                and self.ops.convlstm1_dx_from_supported(self.cin, self.F, c0))

    def _gates_x(self, x, T):
        """fp32, n_timesteps > 1: the input part of the gates for all timesteps in one launch."""
        o, F = self.ops, self.F
        if hasattr(o, "convlstm_gates_x_supported") and o.convlstm_gates_x_supported(x, self.gates, self.cin, F):
            # (the 5 -> 16 layer: 45-row reduction on the matrix pipe with the weights in registers; the 2 -> 2 layer: one pixel per
            # thread on the vector unit — csrc/convlstm1.hip)
            o.convlstm_gates_x(x, self.wx.value, self.b.value, self.gates, self.cin, F)
        else:
            o.conv_fwd(x, self.pkx, self.b.value, self.gates, self.g, act=False)

    def _time_loop_fwd(self, h, B, T):
        """fp32 recurrence of this layer alone (forward_pair: both discriminator layers together)."""
        o, F = self.ops, self.F
        step1 = hasattr(o, "convlstm_step_supported") and o.convlstm_step_supported(h[:B], self.gates[:B], self.pkh, self.g, F)

        def time_loop():
            for t in range(T):
                sl = slice(t * B, (t + 1) * B)
                if t > 0:
                    pv = slice((t - 1) * B, t * B)
                    if step1:
                        o.convlstm_step(h[pv], self.pkh, self.gates[sl], self.c[pv], self.c[sl], h[sl], self.g, F)
                        continue
                    o.conv_fwd(h[pv], self.pkh, None, self.gates[sl], self.g, act=False, accumulate=True)
                    o.lstm_fwd(v2(self.gates[sl]), v2(self.c[pv]), v2(self.c[sl]), v2(h[sl]), F)
                else:
                    o.lstm_fwd(v2(self.gates[sl]), None, v2(self.c[sl]), v2(h[sl]), F)

        packs = ()
        if step1 and hasattr(o, "convlstm_step_prepare"):
            packs = o.convlstm_step_prepare(h[:B], self.pkh, self.gates[:B], self.g, F) or ()
        if T > 2 and hasattr(o, "chain"):
            o.chain(("lstm_fwd", h.data_ptr(), tuple(h.shape), self.gates.data_ptr(), self.c.data_ptr(), self.pkh.wF.data_ptr(),
                     B, T, False, "bf16", bool(step1)) + tuple(packs), time_loop, graphs=self._chain_graphs)
        else:
            time_loop()

    def backward(self, x, h, dh, dx, B, T, need_wgrad, accumulate_dx=False, x2=None, dx_c0=0):
        """dh: total gradient w.r.t. every h_t (modified in place by the BPTT recursion);
        dx: view receiving the input gradient (None to skip).  x2: as in forward.
        dx_c0 > 0 (dx_from_ok): dx[..., :cin - dx_c0] receives the gradient of input channels [dx_c0, cin) only."""
        o, F = self.ops, self.F
        N, H, W, _ = h.shape
        if dx_c0:
            assert self.dx_from_ok(T, dx_c0, need_wgrad, x2) and dx is not None
            o.convlstm1_bwd(x, self.wx.value, self.b.value, dh, None, dx, self.cin, F, accumulate_dx=accumulate_dx, dx_c0=dx_c0)
            return
        if self._fused1(T):
            # input gradient and (need_wgrad) kernel + bias gradient in ONE kernel: the gates are recomputed from x and
            # the dense dgates tensor is never materialised
            kw = {} if x2 is None else {"x2": x2}
            o.convlstm1_bwd(x, self.wx.value, self.b.value, dh, None, dx, self.cin, F, accumulate_dx=accumulate_dx,
                            dw=self.wx.grad if need_wgrad else None, dbias=self.b.grad if need_wgrad else None, **kw)
            return
        assert x2 is None
        self._bwd_buffers(h, B)
        self._time_loop_bwd(h, dh, B, T)
        self._bwd_tail(x, h, dx, B, T, need_wgrad, accumulate_dx)

    def _bwd_buffers(self, h, B):
        N, H, W, _ = h.shape
        if self.dgates is None:
            self.dgates = self.ops.empty(N, H, W, 4 * self.F)
            self.dc = [self.ops.empty(B, H, W, self.F), self.ops.empty(B, H, W, self.F)]

    def _time_loop_bwd(self, h, dh, B, T):
        """BPTT recursion of this layer alone (convlstm_pair_backward: both discriminator layers together)."""
        o, F = self.ops, self.F
        # one launch per timestep where the halo-tile kernel runs the recurrent data gradient (the discriminator's ConvLSTMs):
        # dh_{t-1} += conv_transpose(dgates_t) and, in the same epilogue, the cell backward of timestep t-1
        bstep = T > 1 and hasattr(o, "convlstm_bwd_step_supported") and \
            o.convlstm_bwd_step_supported(dh[:B], self.dgates[:B], self.pkh, self.g, F)
        def time_loop():
            dc_in = None
            for t in range(T - 1, -1, -1):
                sl = slice(t * B, (t + 1) * B)
                pv = slice((t - 1) * B, t * B)
                dc_out = self.dc[t & 1] if t > 0 else None
                if not (bstep and t < T - 1):          # (fused mode: the previous iteration's launch already did this cell backward)
                    o.lstm_bwd(v2(self.gates[sl]), v2(self.c[pv]) if t > 0 else None, v2(self.c[sl]), v2(dh[sl]),
                               v2(dc_in) if dc_in is not None else None, v2(self.dgates[sl]),
                               v2(dc_out) if dc_out is not None else None, F)
                if t > 0:
                    if bstep:
                        pp = slice((t - 2) * B, (t - 1) * B)
                        o.convlstm_bwd_step(self.dgates[sl], self.pkh, dh[pv], self.gates[pv], self.c[pp] if t > 1 else None,
                                            self.c[pv], dc_out, self.dgates[pv], self.dc[(t - 1) & 1] if t > 1 else None, self.g, F)
                    else:
                        o.conv_dgrad(self.dgates[sl], self.pkh, dh[pv], self.g, accumulate=True)
                dc_in = dc_out

        packs = ()
        if bstep and hasattr(o, "convlstm_step_prepare"):
            packs = o.convlstm_step_prepare(h[:B], self.pkh, self.gates[:B], self.g, F) or ()   # (weight layouts current before a replayed loop)
        if T > 2 and hasattr(o, "chain"):
            o.chain(("lstm_bwd", h.data_ptr(), tuple(h.shape), dh.data_ptr(), self.gates.data_ptr(), self.c.data_ptr(),
                     self.dgates.data_ptr(), self.dc[0].data_ptr(), self.dc[1].data_ptr(), self.pkh.wD.data_ptr(), B, T, bool(bstep))
                    + tuple(packs), time_loop, graphs=self._chain_graphs)
        else:
            time_loop()

    def _bwd_tail(self, x, h, dx, B, T, need_wgrad, accumulate_dx):
        """Weight gradients and the input gradient from the dense dgates tensor."""
        o, F = self.ops, self.F
        # n_timesteps = 1: dgates of the forget gate = dc * c_0 * hs' = 0 -> its quarter of the weight gradient and of the data
        # gradient's reduction is skipped (channel ranges [0, F) and [2F, 4F) of the gate tensor: HipOps.conv_dgrad_slice)
        live = T == 1 and F % 4 == 0 and getattr(o, "supports_weight_slices", False) \
            and os.environ.get("WDG_LSTM_LIVE_GATES", "1") != "0"
        if live and hasattr(o, "weight_slices_ok"):
            # (both range widths: F for the input gate, 2F for candidate + output gate; layers that would run on the halo / thin
            # kernels cannot be sliced and keep the full-width calls below)
            live = o.weight_slices_ok(x, self.dgates[..., :F], self.pkx, self.g) and o.weight_slices_ok(x, self.dgates[..., 2 * F:], self.pkx, self.g)
        if need_wgrad:
            def weight_grads():
                # (the bias gradient — the column sums of dgates, 453 MB at batch 8 x T 24 — rides on the input kernel's weight
                # gradient where that kernel has a spare constant-1 row: the discriminator's 5- and 2-channel layers)
                if live:
                    o.conv_wgrad_slice(x, self.dgates[..., :F], self.pkx, 0, F, self.wx.grad, self.g, accumulate=True)
                    o.conv_wgrad_slice(x, self.dgates[..., 2 * F:], self.pkx, 2 * F, 4 * F, self.wx.grad, self.g, accumulate=True)
                    o.colsum(v2(self.dgates), self.b.grad, accumulate=True)
                    return
                o.conv_wgrad(x, self.dgates, self.pkx, self.wx.grad, self.g, accumulate=True, dbias=self.b.grad)
                if T > 1:
                    o.conv_wgrad(h[:(T - 1) * B], self.dgates[B:], self.pkh, self.wh.grad, self.g, accumulate=True)
            joins = getattr(self.net, "_bwd_joins", None)
            if joins is not None:
                self.net._wgrad(weight_grads, joins)      # under the input gradient below (the network's pass joins)
            else:
                weight_grads()
        if dx is not None and live:
            o.conv_dgrad_slice(self.dgates[..., :F], self.pkx, 0, F, dx, self.g, accumulate=accumulate_dx)
            o.conv_dgrad_slice(self.dgates[..., 2 * F:], self.pkx, 2 * F, 4 * F, dx, self.g, accumulate=True)
        elif dx is not None and hasattr(o, "convlstm_gates_dx_supported") and o.convlstm_gates_dx_supported(self.dgates, dx, self.cin, F):
            # (the 2 -> 2-feature layer: one pixel per thread on the vector unit, csrc/convlstm1.hip)
            o.convlstm_gates_dx(self.dgates, self.wx.value, dx, self.cin, F, accumulate=accumulate_dx)
        elif dx is not None:
            o.conv_dgrad(self.dgates, self.pkx, dx, self.g, accumulate=accumulate_dx)


def convlstm_pair_ok(la, lb, ha, hb, B, T):
    """Can the recurrences of the two-feature layer `la` and the 16-feature layer `lb` (the discriminator's two ConvLSTM2D,
    models.py:93,101: independent chains) share their per-timestep launches?  (HipOps.convlstm_pair_step / _bwd_step)"""
    o = la.ops
    if T < 2 or la.F != 2 or lb.F != 16 or not hasattr(o, "convlstm_pair_supported") or la._shape is None or lb._shape is None:
        return False
    return o.convlstm_pair_supported(hb[:B], lb.gates[:B], lb.pkh, ha[:B], la.gates[:B], la.pkh, lb.g)


def convlstm_pair_forward(la, xa, ha, lb, xb, hb, B, T):
    """ConvLSTM.forward of both layers with ONE launch per timestep for the two recurrences (fp32, n_timesteps > 1).  Falls back to
    the layers' own loops where the joint step is not available."""
    o = la.ops
    for l, x, h in ((la, xa, ha), (lb, xb, hb)):
        l._buffers(h.shape[0], h.shape[1], h.shape[2])
        l._gates_x(x, T)
    if not convlstm_pair_ok(la, lb, ha, hb, B, T):
        la._time_loop_fwd(ha, B, T)
        lb._time_loop_fwd(hb, B, T)
        return

    def time_loop():
        for t in range(T):
            sl = slice(t * B, (t + 1) * B)
            if t == 0:
                o.lstm_fwd(v2(la.gates[sl]), None, v2(la.c[sl]), v2(ha[sl]), la.F)
                o.lstm_fwd(v2(lb.gates[sl]), None, v2(lb.c[sl]), v2(hb[sl]), lb.F)
                continue
            pv = slice((t - 1) * B, t * B)
            o.convlstm_pair_step((hb[pv], lb.pkh, lb.gates[sl], lb.c[pv], lb.c[sl], hb[sl]),
                                 (ha[pv], la.pkh, la.gates[sl], la.c[pv], la.c[sl], ha[sl]), lb.g)

    packs = o.convlstm_step_prepare(hb[:B], lb.pkh, lb.gates[:B], lb.g, lb.F) or ()      # (weight layouts current before a replayed loop)
    if T > 2 and hasattr(o, "chain"):
        o.chain(("lstm_pair_fwd", ha.data_ptr(), hb.data_ptr(), tuple(hb.shape), la.gates.data_ptr(), lb.gates.data_ptr(), la.c.data_ptr(),
                 lb.c.data_ptr(), la.pkh.wF.data_ptr(), B, T) + tuple(packs), time_loop, graphs=lb._chain_graphs)
    else:
        time_loop()


def convlstm_pair_backward(la, xa, ha, dha, dxa, lb, xb, hb, dhb, dxb, B, T, need_wgrad, accumulate_dxb=False):
    """ConvLSTM.backward of both layers with one launch per timestep for the two BPTT recursions; dxa / dxb: the views that
    receive the input gradients (None to skip)."""
    o = la.ops
    if not convlstm_pair_ok(la, lb, ha, hb, B, T):
        la.backward(xa, ha, dha, dxa, B, T, need_wgrad)
        lb.backward(xb, hb, dhb, dxb, B, T, need_wgrad, accumulate_dx=accumulate_dxb)
        return
    la._bwd_buffers(ha, B)
    lb._bwd_buffers(hb, B)

    def time_loop():
        for t in range(T - 1, -1, -1):
            sl, pv, pp = slice(t * B, (t + 1) * B), slice((t - 1) * B, t * B), slice((t - 2) * B, (t - 1) * B)
            if t == T - 1:          # (every later cell backward is done by the previous iteration's joint launch)
                for l, dh in ((la, dha), (lb, dhb)):
                    o.lstm_bwd(v2(l.gates[sl]), v2(l.c[pv]) if t > 0 else None, v2(l.c[sl]), v2(dh[sl]), None, v2(l.dgates[sl]),
                               v2(l.dc[t & 1]) if t > 0 else None, l.F)
            if t > 0:
                args = []
                for l, dh in ((lb, dhb), (la, dha)):
                    args.append((l.dgates[sl], l.pkh, dh[pv], l.gates[pv], l.c[pp] if t > 1 else None, l.c[pv], l.dc[t & 1], l.dgates[pv],
                                 l.dc[(t - 1) & 1] if t > 1 else None))
                o.convlstm_pair_bwd_step(args[0], args[1], lb.g)

    packs = o.convlstm_step_prepare(hb[:B], lb.pkh, lb.gates[:B], lb.g, lb.F) or ()
    if T > 2 and hasattr(o, "chain"):
        o.chain(("lstm_pair_bwd", ha.data_ptr(), hb.data_ptr(), tuple(hb.shape), dha.data_ptr(), dhb.data_ptr(), la.gates.data_ptr(),
                 lb.gates.data_ptr(), la.c.data_ptr(), lb.c.data_ptr(), la.dgates.data_ptr(), lb.dgates.data_ptr(), la.dc[0].data_ptr(),
                 la.dc[1].data_ptr(), lb.dc[0].data_ptr(), lb.dc[1].data_ptr(), la.pkh.wD.data_ptr(), B, T) + tuple(packs), time_loop,
                graphs=lb._chain_graphs)
    else:
        time_loop()
    la._bwd_tail(xa, ha, dxa, B, T, need_wgrad, False)
    lb._bwd_tail(xb, hb, dxb, B, T, need_wgrad, accumulate_dxb)
