"""Generator and discriminator graphs of the reference, as explicit forward/backward programs.

Follows /root/reference/src/downscaling/gan/models.py line by line (citations inline).  Internal
activations are TIME-MAJOR channels-last: image index n = t*B + b, shape [T*B, H, W, C]; channel
concatenations are zero-copy views into wider buffers; channel counts that are not multiples of 4
are zero-padded (input 23 -> 24, wind 2 -> 4, low+high 5 -> 8).
"""
import os

import numpy as np

from .common import ConvGeom, round4, v2
from .layers import (LN_EPS, LRELU, BatchNorm, Conv, ConvLSTM, Dense, LayerNorm, convlstm_pair_backward,
                     convlstm_pair_forward)
from .params import ParamStore, glorot_uniform, zeros_init


class _Net:
    def __init__(self, ops, sync=None):
        self.ops = ops
        self.sync = sync
        self.params = ParamStore(ops)
        self._layers = []
        self._bufs = {}
        self._prep = None
        self._packed_version = -1

    def _add(self, layer):
        self._layers.append(layer)
        return layer

    def _finalize(self, seed):
        self.params.finalize(np.random.default_rng(seed))
        for l in self._layers:
            if hasattr(l, "build"):
                l.build()
        self._packed_version = self.params.version

    def _prepare(self, training):
        """What a Keras call does before the first layer runs: SN power iteration of every wrapped layer
        (training only: w <- w / sigma and u in place, tfa SpectralNormalization) and refresh of the
        kernel-layout weight copies that are stale — batched over the whole network."""
        if self._prep is None:
            entries = []
            for l in self._layers:
                if hasattr(l, "prep_entries"):
                    entries += l.prep_entries()
            self._prep = self.ops.make_prep_batch(entries)
        self._prep.run(sn=training, pack_all=self._packed_version != self.params.version)
        self._packed_version = self.params.version

    # ---- weight gradients off the critical path ---------------------------------------------------
    # History: round 3 on (70.9 -> 69.75 ms), round 4 off (with the generator and the twin discriminator on streams of their own a
    # fourth stream cost more than its tails saved: 66.1-66.7 ms without against 66.9 with, profiles/r04e_queues.txt), round 5 ON
    # again: once the LayerNorm backward moved into the data gradients and the second stages got short, the weight gradients are
    # what is left to run beside the data-gradient chain — 64.2 / 64.0 -> 63.6 / 63.3 ms, same box, alternating
    # (profiles/r05x_ab_step_wgrad_stream.txt).  WDG_WGRAD_STREAM=0 disables.
    # (parsed as documented: 0 = off, 2 = only the small ones, anything else = on)
    wgrad_stream = os.environ.get("WDG_WGRAD_STREAM", "1") not in ("0", "2")
    # WDG_WGRAD_STREAM=2: only the weight gradients flagged `small` (the discriminator's 27 x 27 / 8 x 8 / 2 x 2 blocks: launches of
    # 25-45 TFLOP/s that leave most of the chip idle beside an equally small data-gradient chain)
    wgrad_stream_small = os.environ.get("WDG_WGRAD_STREAM", "1") == "2"

    def _wgrad(self, fn, joins, small=False):
        """Runs `fn` (the weight-gradient launches of one layer) on the "wgrad" side stream, after everything enqueued so far
        (its operands are ready), while the caller goes on with the data gradient on the current stream: the two read the
        same dz and write different buffers, and nothing needs dW before the optimizer step.  On the small maps of the
        discriminator's stack and at T > 1 neither kernel fills the chip, and everywhere the one's tail runs under the
        other.  `joins` collects the forks; the pass joins them before it returns (the next pass overwrites the
        activations and gradient buffers the weight gradients read)."""
        if not (self.wgrad_stream or (small and self.wgrad_stream_small)):
            fn()
            return
        with self.ops.fork("wgrad", stream=getattr(self, "wgrad_side", None)) as side:
            fn()
        joins.append(side)

    @staticmethod
    def _join(joins):
        for side in joins[-1:]:          # one stream: its last fork's join covers the earlier ones
            side.join()
        joins.clear()

    # ---- [B,T,...] <-> time-major ---------------------------------------------------------------
    def to_time_major(self, src, dst):
        """src [B,T,H,W,C] (API layout) -> dst[..., :C] of a [T*B,H,W,C'] buffer."""
        B, T = src.shape[0], src.shape[1]
        C = src.shape[-1]
        if T == 1:
            self.ops.copy_channels(src[:, 0], dst[..., :C])
            return
        # one strided copy for the whole (B,T) -> (T,B) permutation instead of T launches (72 tiny kernels per
        # generator forward at the shipped T = 24: 9 % of the bf16 inference forward)
        self.ops.permute_bt(src, dst.view(T, B, *dst.shape[1:]))

    def from_time_major(self, src, dst):
        """src [T*B,H,W,>=C] -> dst [B,T,H,W,C]."""
        B, T = dst.shape[0], dst.shape[1]
        C = dst.shape[-1]
        if T == 1:
            self.ops.copy_channels(src[..., :C], dst[:, 0])
            return
        self.ops.permute_bt(src.view(T, B, *src.shape[1:]), dst)


class GeneratorNet(_Net):
    """make_generator, gan/models.py:9-73."""

    def __init__(self, ops, image_size, in_channels, noise_channels, out_channels, n_timesteps,
                 feature_channels=128, seed=0, sync=None):
        super().__init__(ops, sync)
        assert image_size % 4 == 0                      # models.py:19
        assert feature_channels % 8 == 0                # models.py:20
        S, F = image_size, feature_channels
        cin = in_channels + noise_channels              # models.py:21
        IF = cin * 8 if cin * 8 <= F else F             # models.py:31
        if F / 8 < out_channels:
            # models.py:66-68: that branch skips the upsampling and fails its own shape assertion
            raise AssertionError("feature_channels / 8 must be >= out_channels (reference else-branch is dead)")
        # feature_channels / 4 is the width of the first segment of the [conv-transpose path | res_2] concatenation
        # (models.py:60); the skip tensor behind it is addressed in place and needs a 16-byte aligned start, so for
        # feature_channels % 16 == 8 the segment is followed by zero alignment channels and the kernel of the layer that
        # reads the concatenation carries matching zero rows (params.Var.gap) — every % 8 width of the reference builds
        self.F4p = round4(F // 4)
        self.S, self.F, self.IF, self.T = S, F, IF, n_timesteps
        self.cin, self.in_channels, self.noise_channels, self.out_channels = cin, in_channels, noise_channels, out_channels
        L = "layer_with_weights-"
        self.c0 = self._add(Conv(self, L + "0", 8, cin, IF, 2, 3, sn=True))           # :32-33
        self.bn1 = self._add(BatchNorm(self, L + "1", IF))                            # :34
        self.c2 = self._add(Conv(self, L + "2", 4, IF, F, 2, 1, sn=True))             # :38-39
        self.bn3 = self._add(BatchNorm(self, L + "3", F))                             # :40
        self.lstm = self._add(ConvLSTM(self, L + "4", F, F))                          # :45
        self.c5 = self._add(Conv(self, L + "5", 3, F, F // 2, 1, 1, sn=True))         # :49
        self.bn6 = self._add(BatchNorm(self, L + "6", F // 2))                        # :50
        self.c7 = self._add(Conv(self, L + "7", 2, F // 4, F // 2 + F, 2, 0, transposed=True, sn=True))  # :54-55
        self.bn8 = self._add(BatchNorm(self, L + "8", F // 4))                        # :56
        self.c9 = self._add(Conv(self, L + "9", 5, F // 8, F // 4 + IF, 1, 2, transposed=True,            # :60-64
                                 cout_gap=(F // 4, self.F4p - F // 4) if self.F4p != F // 4 else None))
        self.bn10 = self._add(BatchNorm(self, L + "10", F // 8))                      # :69
        self.c11 = self._add(Conv(self, L + "11", 3, F // 8, out_channels, 1, 1, act=False))            # :70-71
        self._finalize(seed)

    def buffers(self, B):
        b = self._bufs.get(B)
        if b is not None:
            return b
        o, S, F, IF, T = self.ops, self.S, self.F, self.IF, self.T
        N, S2, S4 = T * B, self.S // 2, self.S // 4
        b = dict(
            x0=o.zeros(N, S, S, round4(self.cin)),
            y0=o.empty(N, S2, S2, IF),
            cat2=o.zeros(N, S2, S2, self.F4p + IF),    # [c7 path (+ zero alignment channels) | res_2]
            y2=o.empty(N, S4, S4, F),
            cat4=o.empty(N, S4, S4, F // 2 + F),       # [c5 path | res_4]
            h=o.zeros(N, S4, S4, F),
            y5=o.empty(N, S4, S4, F // 2),
            y7=o.zeros(N, S2, S2, self.F4p),
            y9=o.zeros(N, S, S, round4(F // 8)),       # (F / 8 not a multiple of 4: zero pad channels)
            z9=o.zeros(N, S, S, round4(F // 8)),
            out=o.zeros(N, S, S, round4(self.out_channels)),
        )
        self._bufs = {B: b}      # one resident batch size at a time
        self._grad_bufs = None
        return b

    def grad_buffers(self, B):
        if self._grad_bufs is None:
            o, S, F, IF, T = self.ops, self.S, self.F, self.IF, self.T
            N, S2, S4 = T * B, S // 2, S // 4
            self._grad_bufs = dict(
                dz9=o.zeros(N, S, S, round4(F // 8)),
                dcat2=o.zeros(N, S2, S2, self.F4p + IF),
                dcat4=o.empty(N, S4, S4, F // 2 + F),
                dh=o.zeros(N, S4, S4, F),
            )
        return self._grad_bufs

    @staticmethod
    def _scratch_pool(b):
        """Operator scratch (the column tensor of the upsample + transposed-conv block) owned by THIS buffer set, like the
        graphs captured on it: see HipOps._scratch."""
        return b.setdefault("scratch", {})

    # ---- inputs ------------------------------------------------------------------------------------
    def set_image(self, image):
        """image [B,T,S,S,in] -> channels [0:in] of the concatenated input buffer (models.py:28)."""
        b = self.buffers(image.shape[0])
        b["x0_src"] = None
        self.to_time_major(image, b["x0"])

    def set_noise(self, noise):
        b = self.buffers(noise.shape[0])
        B, T = noise.shape[0], noise.shape[1]
        if T == 1:
            self.ops.copy_channels(noise[:, 0], b["x0"][..., self.in_channels:self.cin])
        else:
            self.ops.permute_bt(noise, b["x0"].view(T, B, *b["x0"].shape[1:])[..., self.in_channels:self.cin])

    def input_rows(self, B):
        """[T*B*S*S, ld] view of the whole input buffer ([image | noise | zero alignment channels] per pixel, time-major rows)."""
        self.buffers(B)["x0_src"] = None
        return v2(self.buffers(B)["x0"])

    def input_rows16(self, B, fmt):
        """The same view of the input buffer in the 16-bit operand format `fmt` — what the first layer of the inference-precision
        forward rounds the input to while staging; a caller that assembles the input on the device (HipOps.input_assemble) can
        write it there directly.  None when that forward would not read it (no 16-bit first layer for this shape / WDG_ACT16=0).
        The caller marks the buffer as the current input with `mark_input16`."""
        b = self.buffers(B)
        key = "x0_" + fmt
        if key not in b:
            b[key] = None
            ok = getattr(self.ops, "act16_conv_ok", None)
            x0 = b["x0"]
            if ok is not None and self.ops.act16 and x0.shape[3] % 8 == 0 and os.environ.get("WDG_ACT16_INPUT", "1") != "0":
                x16 = self.ops.zeros(*x0.shape, dtype=self.ops.H16_DTYPES[fmt])
                if ok(x16, b["cat2"][..., self.F4p:], self.c0.pk, self.c0.g, False, 1, 0) and \
                        ok(x16, b["cat2"][..., self.F4p:], self.c0.pk, self.c0.g, False, 1, 1):
                    b[key] = x16
        return v2(b[key]) if b[key] is not None and self.ops.act16 else None

    def mark_input16(self, B, fmt):
        self.buffers(B)["x0_src"] = fmt

    def noise_view(self, B):
        """[T*B*S*S, noise_channels] view of the input buffer: noise can be generated in place."""
        self.buffers(B)["x0_src"] = None
        return v2(self.buffers(B)["x0"][..., self.in_channels:self.cin])

    # ---- inference forward as a HIP graph ----------------------------------------------------------------
    def forward_inference(self, B, precision="fp32"):
        """The inference forward replayed from a captured HIP graph: at the shipped sequence length (T = 24) a forward
        is ~330 launches of mostly small per-timestep kernels (7 ms of GPU time in the 16-bit path) and the Python /
        ctypes launch cost per call is of the same order.  The graph is captured once per (batch, precision, weights
        version) after an eager warm-up (plans, scratch, packed / 16-bit weights all exist by then); inputs and output
        live in the network's resident buffers, so callers write x0 (set_image / set_noise / noise_view) and read the
        returned buffer exactly as with forward().  Falls back to the eager path if capture is unavailable."""
        # graphs live WITH the buffer set they were captured on: a different batch size replaces the buffers (one
        # resident size at a time) and must drop the graphs that hold their addresses
        graphs = self.buffers(B).setdefault("graphs", {})
        self._graphs = graphs
        if not getattr(self.ops, "supports_graphs", False) or getattr(self, "_graphs_disabled", False):
            return self.forward(B, training=False, precision=precision)
        # weights version + prep epoch: a training-mode forward (SN power iteration, no optimizer step) rewrites w and the
        # packed fp32 layouts in place, but the 16-bit copies / composite kernels are reconverted lazily on the HOST path
        # only — a graph captured before must not be replayed on them
        key = (B, precision, self.params.version, self._prep.epoch if self._prep is not None else 0, self.buffers(B).get("x0_src"))
        entry = graphs.get(key)
        if entry is None:
            # capture pays off only for repeated calls: the first two forwards of a configuration run eagerly (they also
            # create plans, scratch and the packed / 16-bit weights), the third is captured
            seen = graphs.setdefault("seen", {})
            seen[key] = seen.get(key, 0) + 1
            if seen[key] <= 2:
                return self.forward(B, training=False, precision=precision)
            import torch
            try:
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = self.forward(B, training=False, precision=precision)
            except Exception:                                             # capture not possible here: stay eager
                torch.cuda.synchronize()
                self._graphs_disabled = True
                return self.forward(B, training=False, precision=precision)
            for k in [k for k in graphs if isinstance(k, tuple) and k[2:] != key[2:]]:
                del graphs[k]                                             # graphs of older weights
            seen.clear()
            entry = graphs[key] = (graph, out)
        entry[0].replay()
        return entry[1]

    # ---- forward -----------------------------------------------------------------------------------
    def forward(self, B, training, need_backward=None, precision="fp32"):
        """Runs on the resident input buffer; returns the [T*B,S,S,round4(out)] time-major output.
        precision="bf16" / "fp16" (inference only): the implicit-GEMM layers use 16-bit MFMA operands with fp32
        accumulation and fuse the inference BatchNorm into their epilogue.
        need_backward is kept for interface stability: no forward materialises the bilinear-upsampled tensor any
        more (the backward of the upsample + transposed-conv block works on the low-res grid, ops.upconv_bwd)."""
        if need_backward is None:
            need_backward = training
        if precision in ("bf16", "fp16") and training:       # (before anything is prepared: a refused call leaves no trace)
            raise ValueError("the 16-bit operand path is inference-only")
        b = self.buffers(B)
        F, IF, T = self.F, self.IF, self.T
        self._prepare(training)
        res2 = b["cat2"][..., self.F4p:]
        res4 = b["cat4"][..., F // 2:]
        if precision in ("bf16", "fp16"):
            f = precision
            # Activations whose every reader is a 16-bit layer are stored in the operand format by their producers (the readers
            # would round them to it while staging: the same bits, half the bytes): the [c7 | c0] concatenation that feeds c2 and
            # the upsample block — the largest tensor of the forward — and (below) the 16 channels between the last two layers.
            cat2 = b["cat2"]
            key = "cat2_" + f
            ok = getattr(self.ops, "act16_conv_ok", None)
            if key not in b:
                b[key] = None
                if ok is not None and self.ops.act16 and cat2.shape[3] % 8 == 0 and self.F4p % 8 == 0:
                    c16 = self.ops.zeros(*cat2.shape, dtype=self.ops.H16_DTYPES[f])
                    if ok(b["x0"], c16[..., self.F4p:], self.c0.pk, self.c0.g, False, 0, 1) and \
                            ok(c16[..., self.F4p:], res4, self.c2.pk, self.c2.g, False, 1, 0) and \
                            ok(b["cat4"], c16[..., :self.F4p], self.c7.pk, self.c7.g, True, 0, 1) and \
                            self.ops.act16_upconv_in_ok(c16, self.c9.pk):
                        b[key] = c16
            if b[key] is not None and self.ops.act16:
                cat2 = b[key]
                res2 = cat2[..., self.F4p:]
            # ... and the quarter-resolution tensors: the [c5 | c2] concatenation (read by the ConvLSTM's gate convolution and by c7)
            # and the ConvLSTM's hidden state (read by its next step and by c5) — every reader a 16-bit layer
            cat4, hbuf = b["cat4"], b["h"]
            key4 = "cat4_" + f
            if key4 not in b:
                b[key4] = b["h_" + f] = None
                okl = getattr(self.ops, "act16_lstm_ok", None)
                if ok is None:
                    ok = getattr(self.ops, "act16_conv_ok", None)
                if ok is not None and okl is not None and self.ops.act16 and cat4.shape[3] % 8 == 0 and (F // 2) % 8 == 0 and F % 8 == 0 \
                        and T > 1 and os.environ.get("WDG_ACT16_QUARTER", "1") != "0":
                    c16 = self.ops.zeros(*cat4.shape, dtype=self.ops.H16_DTYPES[f])
                    h16 = self.ops.zeros(*hbuf.shape, dtype=self.ops.H16_DTYPES[f])
                    self.lstm._buffers(hbuf.shape[0], hbuf.shape[1], hbuf.shape[2])
                    if ok(res2, c16[..., F // 2:], self.c2.pk, self.c2.g, False, int(res2.dtype != self.ops.dtype), 1) and \
                            okl(c16[..., F // 2:], self.lstm.gates, self.lstm.pkx, self.lstm.pkh, self.lstm.g, F) and \
                            ok(h16, c16[..., :F // 2], self.c5.pk, self.c5.g, False, 1, 1) and \
                            ok(c16, cat2[..., :self.F4p], self.c7.pk, self.c7.g, True, 1, int(cat2.dtype != self.ops.dtype)):
                        b[key4], b["h_" + f] = c16, h16
            # (decided per call: wdg_set_tuning switches such as lstm16_fused / patch_h16 move the ConvLSTM off the route that keeps
            # its state in the operand format)
            if b[key4] is not None and self.ops.act16 and \
                    self.ops.act16_lstm_ok(b[key4][..., F // 2:], self.lstm.gates, self.lstm.pkx, self.lstm.pkh, self.lstm.g, F):
                cat4, hbuf = b[key4], b["h_" + f]
                res4 = cat4[..., F // 2:]
            # the input itself, when its producer wrote it in the operand format (input_rows16 / mark_input16)
            x0 = b["x0_" + f] if b.get("x0_src") == f and b.get("x0_" + f) is not None and self.ops.act16 else b["x0"]
            self.c0.forward_bf16(x0, res2, affine=self.bn1.infer_affine(), fmt=f)
            self.c2.forward_bf16(res2, res4, affine=self.bn3.infer_affine(), fmt=f)
            self.lstm.forward(res4, hbuf, B, T, bf16=True, fmt=f)
            self.c5.forward_bf16(hbuf, cat4[..., :F // 2], affine=self.bn6.infer_affine(), fmt=f)
            self.c7.forward_bf16(cat4, cat2[..., :self.F4p], affine=self.bn8.infer_affine(), fmt=f)
            z9 = b["z9"]
            ok16 = getattr(self.ops, "act16_output_conv_ok", None)
            aff10 = self.bn10.infer_affine()
            if ok16 is not None and z9.shape[3] == 16 and ok16(self.c9.pk, self.c11.pk, self.c11.g, x_low=cat2,
                                                               bias=self.c9.b.value, affine=aff10):
                # the 16-channel activation between the last two layers in the operand format: its only reader rounds to that
                # format anyway — the same values, half the bytes of the largest-by-pixels tensor of the forward
                key = "z9_" + f
                if key not in b or b[key].shape != z9.shape:
                    b[key] = self.ops.zeros(*z9.shape, dtype=self.ops.H16_DTYPES[f])
                z9 = b[key]
            self.ops.upconv_fwd_bf16(cat2, self.c9.pk, self.c9.b.value, z9, self.c9.g, act=True,
                                     affine=aff10, fmt=f, pool=self._scratch_pool(b))
            self.ops.conv_halo_fwd_bf16(z9, self.c11.pk, self.c11.b.value, b["out"], self.c11.g, act=False, fmt=f)
            return b["out"]
        if precision != "fp32":
            raise ValueError(f"unknown precision {precision!r}")
        if not training:
            # inference: every BatchNormalization is a per-channel affine of moving statistics, applied by the epilogue of
            # the launch that produces its input (no pre-norm tensor, no normalisation pass)
            self.c0.forward(b["x0"], res2, bn_affine=self.bn1.infer_affine())
            self.c2.forward(res2, res4, bn_affine=self.bn3.infer_affine())
            self.lstm.forward(res4, b["h"], B, T)
            self.c5.forward(b["h"], b["cat4"][..., :F // 2], bn_affine=self.bn6.infer_affine())
            self.c7.forward(b["cat4"], b["cat2"][..., :self.F4p], bn_affine=self.bn8.infer_affine())
            self.ops.upconv_fwd(b["cat2"], self.c9.pk, self.c9.b.value, b["z9"], self.c9.g, act=True,
                                pool=self._scratch_pool(b), bn_affine=self.bn10.infer_affine())
            self.c11.forward(b["z9"], b["out"])
            return b["out"]
        # training: batch statistics come out of the producing launch's epilogue (BatchNorm.begin_stats), the
        # normalisation itself is one pass (bn_apply) once they are complete
        self.c0.forward(b["x0"], b["y0"], bn_stats=self.bn1.begin_stats())
        self.bn1.forward(v2(b["y0"]), v2(res2), True, have_stats=True)
        self.c2.forward(res2, b["y2"], bn_stats=self.bn3.begin_stats())
        self.bn3.forward(v2(b["y2"]), v2(res4), True, have_stats=True)
        self.lstm.forward(res4, b["h"], B, T)
        self.c5.forward(b["h"], b["y5"], bn_stats=self.bn6.begin_stats())
        self.bn6.forward(v2(b["y5"]), v2(b["cat4"][..., :F // 2]), True, have_stats=True)
        self.c7.forward(b["cat4"], b["y7"], bn_stats=self.bn8.begin_stats())
        self.bn8.forward(v2(b["y7"]), v2(b["cat2"][..., :self.F4p]), True, have_stats=True)
        self.ops.upconv_fwd(b["cat2"], self.c9.pk, self.c9.b.value, b["y9"], self.c9.g, act=True,      # :62-64 fused
                            pool=self._scratch_pool(b), bn_stats=self.bn10.begin_stats())
        self.bn10.forward(v2(b["y9"]), v2(b["z9"]), True, have_stats=True)
        self.c11.forward(b["z9"], b["out"])
        return b["out"]

    # ---- backward ----------------------------------------------------------------------------------
    def backward(self, B, dout):
        """dout: gradient w.r.t. the time-major output [T*B,S,S,round4(out)] (pad channels zero).
        Accumulates into params.grads (generator inputs need no gradient)."""
        b, g = self.buffers(B), self.grad_buffers(B)
        o, F, IF, T = self.ops, self.F, self.IF, self.T
        res2, res4 = b["cat2"][..., self.F4p:], b["cat4"][..., F // 2:]
        # c11 (linear)
        joins = self._bwd_joins = []
        o.colsum(v2(dout[..., :self.out_channels]), self.c11.b.grad, accumulate=True)
        self._wgrad(lambda: self.c11.backward_weights(b["z9"], dout), joins)
        self.c11.backward_input(dout, g["dz9"])
        # bn10 + LeakyReLU of c9
        self.bn10.backward(v2(g["dz9"]), v2(b["y9"]), v2(g["dz9"]), self.c9.b.grad_pad)
        o.upconv_bwd(b["cat2"], g["dz9"], self.c9.pk, self.c9.w.grad, g["dcat2"], self.c9.g,   # :60-64 backward
                     pool=self._scratch_pool(b), **({"wgrad_async": lambda fn: self._wgrad(fn, joins)}
                                                    if getattr(o, "supports_streams", False) else {}))
        # bn8 + c7
        d7 = g["dcat2"][..., :self.F4p]
        self.bn8.backward(v2(d7), v2(b["y7"]), v2(d7), self.c7.b.grad_pad)
        self._wgrad(lambda: self.c7.backward_weights(b["cat4"], d7), joins)
        self.c7.backward_input(d7, g["dcat4"])
        # bn6 + c5
        d5 = g["dcat4"][..., :F // 2]
        self.bn6.backward(v2(d5), v2(b["y5"]), v2(d5), self.c5.b.grad_pad)
        self._wgrad(lambda: self.c5.backward_weights(b["h"], d5), joins)
        self.c5.backward_input(d5, g["dh"])
        # ConvLSTM; its input gradient adds to the skip gradient already in dcat4[..., F/2:]
        dres4 = g["dcat4"][..., F // 2:]
        self.lstm.backward(res4, b["h"], g["dh"], dres4, B, T, need_wgrad=True, accumulate_dx=True)
        # bn3 + c2
        self.bn3.backward(v2(dres4), v2(b["y2"]), v2(dres4), self.c2.b.grad_pad)
        self._wgrad(lambda: self.c2.backward_weights(res2, dres4), joins)
        dres2 = g["dcat2"][..., self.F4p:]
        self.c2.backward_input(dres4, dres2, accumulate=True)
        # bn1 + c0
        self.bn1.backward(v2(dres2), v2(b["y0"]), v2(dres2), self.c0.b.grad_pad)
        self.c0.backward_weights(b["x0"], dres2)
        self._join(joins)


def discriminator_plan(size, channels):
    """The three `while` loops of make_discriminator (models.py:111-136) as a list of conv blocks
    (k, stride, pad, cin, cout, out_size).  `i > 1` (models.py:127) cannot happen."""
    blocks = []
    while size >= 16:                                   # models.py:111
        nxt = (size + 2 - 7) // 3 + 1
        blocks.append((7, 3, 1, channels, channels * 2, nxt))
        size, channels = nxt, channels * 2
    i = 0
    while size >= 4:                                    # models.py:120
        nxt = (size + 2 - 7) // 3 + 1
        if nxt <= 0:
            raise ValueError(f"discriminator: a {size}x{size} map cannot take another 7x7 stride-3 conv")
        blocks.append((7, 3, 1, channels, channels * 2, nxt))
        size, channels = nxt, channels * 2
        i += 1
    if i > 1:
        raise NotImplementedError("shortcut_convolution branch (models.py:127-130) is unreachable in the reference")
    while size > 2:                                     # models.py:132
        nxt = (size - 3) // 2 + 1
        if nxt <= 0:
            raise ValueError("discriminator: map too small for the 3x3 stride-2 conv")
        blocks.append((3, 2, 0, channels, channels * 2, nxt))
        size, channels = nxt, channels * 2
    return blocks, size, channels


def discriminator_shortcut(size, channels):
    """The split connection of models.py:118,127-130 as the shipped weights-55 discriminator was trained with it
    (SURVEY 8 a2 note 2: `layer_with_weights-11/layer/w [6,6,128,256]` exists in the checkpoint although the published
    test `i > 1` can never succeed): taken when the `>= 4` loop ran once.  Returns None or
    (number of blocks before the tap, tap size, tap channels, k, stride, pad, target size, out channels) with the
    geometry of shortcut_convolution (tf_utils.py:15-32)."""
    n_src = 0
    while size >= 16:
        size, channels, n_src = (size + 2 - 7) // 3 + 1, channels * 2, n_src + 1
    if size < 4:
        return None
    target = (size + 2 - 7) // 3 + 1
    if target <= 0:
        return None
    if target == 1:                                                                    # tf_utils.py:18-21
        k, stride, pad = size, 1, 0
    else:
        stride = -(-(2 + size) // (target - 1))                                        # :23 ceil
        pad = -(-(stride * (target - 1) - size) // 2) + 1 + 2                          # :24-25
        k = stride * (1 - target) + size + 2 * pad                                     # :26
    return n_src, size, channels, k, stride, pad, target, channels * 2


class DiscriminatorNet(_Net):
    """make_discriminator, gan/models.py:76-142."""

    def __init__(self, ops, low_res_size, high_res_size, low_res_channels, high_res_channels, n_timesteps,
                 feature_channels=16, seed=1, sync=None, shortcut_variant=False):
        super().__init__(ops, sync)
        self._ctor = dict(low_res_size=low_res_size, high_res_size=high_res_size, low_res_channels=low_res_channels,
                          high_res_channels=high_res_channels, n_timesteps=n_timesteps, feature_channels=feature_channels,
                          seed=seed, shortcut_variant=shortcut_variant)
        self._twin = None
        if low_res_size != high_res_size:               # models.py:89-91
            raise NotImplementedError("The discriminator assumes that the low res and high res images have the "
                                      "same size.Perhaps you should upsample your low res image first?")
        if feature_channels % 4 != 0:
            raise NotImplementedError("this build needs discriminator feature_channels % 4 == 0")
        S, Fd, T = high_res_size, feature_channels, n_timesteps
        self.S, self.Fd, self.T = S, Fd, T
        self.cl, self.ch = low_res_channels, high_res_channels
        L = "layer_with_weights-"
        # checkpoint numbering of weights-55.ckpt/discriminator.index (breadth-first layer order)
        self.lstm_a = self._add(ConvLSTM(self, L + "0", self.ch, self.ch))                    # :93
        self.lstm_b = self._add(ConvLSTM(self, L + "1", self.cl + self.ch, Fd))               # :101
        self.conv_a = self._add(Conv(self, L + "2", 3, self.ch, Fd, 1, 1, sn=True))           # :94-96
        self.conv_b = self._add(Conv(self, L + "3", 3, Fd, Fd, 1, 1, sn=True))                # :102-104
        self.ln_a = self._add(LayerNorm(self, L + "4", Fd))                                   # :97
        self.ln_b = self._add(LayerNorm(self, L + "5", Fd))                                   # :105
        plan, self.final_size, self.final_ch = discriminator_plan(S, 2 * Fd)
        sc = discriminator_shortcut(S, 2 * Fd) if shortcut_variant else None
        self.blocks = []
        self.shortcut = None
        # the two input branches on two streams.  WDG_OVERLAP_BRANCHES: 0 never, 1 (default) at T = 1 only, 2 always, 3 at T > 1
        # only (the round-2 behaviour).  T = 1: worth 0.6 ms per step once the generator runs beside the discriminator (70.95 ->
        # 70.35 ms, same box).  T > 1: two chains of small DEPENDENT launches side by side are slower than one after the other
        # (two 16-feature step chains: 75 us per step pair on two streams, 52 us on one — every kernel boundary of one chain
        # writes back / invalidates the per-XCD L2s under the other's running kernel); 91.6 -> 88.4 ms per T = 24 step without it
        # Round 4: OFF by default at every T — beside the generator / twin streams the branch overlap loses (68.9 vs 67.2 ms at eight
        # hardware queues, neutral at four: profiles/r04f_sched.txt)
        # WDG_MIX_IN_PLACE=1: the high-res channels of concat(low, high) read in place by the fused ConvLSTM kernels (ConvLSTM.x2_ok).
        # Measured neutral and OFF by default: the 16 two-channel copies per step it removes (0.31 ms on one stream) come back as
        # pixel-per-lane 4-byte requests in the kernels' halo staging (fused backward 371 -> 406 us, forward 95 -> 105 us: those
        # stagings are bound by the texture path's line count), 64.26 vs 64.27 ms per step (profiles/r05hij_ab_step_*.txt)
        self.mix_in_place = os.environ.get("WDG_MIX_IN_PLACE", "0") != "0"
        self.chain_ln_bwd = os.environ.get("WDG_CHAIN_LN_BWD", "1") != "0"       # (A/B switch: LayerNorm backward in the upstream data gradient's epilogue)
        mode = os.environ.get("WDG_OVERLAP_BRANCHES", "0")
        self.overlap_branches = mode != "0"
        self.overlap_branches_t1 = mode in ("1", "2")
        self.overlap_branches_tn = mode in ("2", "3")
        idx = 6
        for n, (k, s, p, ci, co, osz) in enumerate(plan):
            conv = self._add(Conv(self, L + str(idx), k, ci, co, s, p, sn=True))              # :113-114,122-123,134
            if sc is not None and n == sc[0]:
                # checkpoint numbering: conv_<size> idx, shortcut_conv idx+1, LN(main) idx+2, LN(shortcut) idx+3
                _, ssz, sci, sk, sstr, spad, tgt, sco = sc
                if tgt == 1:
                    sstr = sk           # a single full-size window: the stride is immaterial
                if sstr < sk:
                    raise NotImplementedError("shortcut_convolution with overlapping windows (stride < kernel)")
                assert (tgt, sco) == (osz, co)
                sconv = self._add(Conv(self, L + str(idx + 1), sk, sci, sco, sstr, spad, sn=True))   # tf_utils.py:27-29
                ln = self._add(LayerNorm(self, L + str(idx + 2), co))
                sln = self._add(LayerNorm(self, L + str(idx + 3), sco))                       # tf_utils.py:31
                self.shortcut = dict(block=n, conv=sconv, ln=sln, size=ssz, cin=sci, k=sk, stride=sstr, pad=spad,
                                     target=tgt, cout=sco)
                idx += 4
            else:
                ln = self._add(LayerNorm(self, L + str(idx + 1), co))                         # :116,125,136
                idx += 2
            self.blocks.append((conv, ln, osz, co))
        K = self.final_size * self.final_size * self.final_ch
        self.K = K
        self.dense_w = self.params.add(L + f"{idx}/layer/kernel", (K, 1), glorot_uniform(K, 1))   # :138
        self.dense_b = self.params.add(L + f"{idx}/layer/bias", (1,), zeros_init)
        self._finalize(seed)

    def buffers(self, B):
        b = self._bufs.get(B)
        if b is not None:
            return b
        o, S, Fd, T = self.ops, self.S, self.Fd, self.T
        N = T * B
        chp = round4(self.ch)
        b = dict(
            hi=o.zeros(N, S, S, chp),                          # high-res input, padded
            mix=o.zeros(N, S, S, round4(self.cl + self.ch)),   # concat(low, high), models.py:100
            ha=o.zeros(N, S, S, chp),
            ya=None if self._fused_conv_ln(self.conv_a) else o.empty(N, S, S, Fd),   # the fused branch keeps no pre-norm tensor
            hb=o.zeros(N, S, S, Fd),
            yb=o.empty(N, S, S, Fd),
            cat=o.empty(N, S, S, 2 * Fd),                      # concat(hr, mix), models.py:108
            ys=[], zs=[],
            score=o.empty(B),
            # gradients
            dcat=o.empty(N, S, S, 2 * Fd),
            dha=o.zeros(N, S, S, chp),
            dhb=o.zeros(N, S, S, Fd),
            dhi=o.zeros(N, S, S, chp),
            dmix=o.zeros(N, S, S, round4(self.cl + self.ch)),
            dhigh=o.zeros(N, S, S, chp),
            dzs=[],
        )
        for (conv, ln, osz, co) in self.blocks:
            b["ys"].append(o.empty(N, osz, osz, co))
            b["zs"].append(o.empty(N, osz, osz, co))
            b["dzs"].append(o.empty(N, osz, osz, co))
        if self.shortcut is not None:
            sc = self.shortcut
            t, kc = sc["target"], sc["k"] * sc["k"] * sc["cin"]
            b.update(sc_patch=o.empty(N, t, t, kc), sc_dpatch=o.empty(N, t, t, kc), sc_y=o.empty(N, t, t, sc["cout"]),
                     sc_z=o.empty(N, t, t, sc["cout"]), sc_dz=o.empty(N, t, t, sc["cout"]))
        self._bufs = {B: b}
        return b

    def twin(self, k=0):
        """A second (k = 0) / third (k = 1) network of the same graph with its OWN variables and activations, for train steps
        whose discriminator loss couples the real and the generated scores (GanEngine._critic_coupled) and for the critic
        schedules that run the real / generated passes beside the gradient-penalty pass (GanEngine._critic_pipelined): it holds
        a pass — the variable values that pass read and its activations — while this network runs another one."""
        if self._twin is None:
            self._twin = {}
        if k not in self._twin:
            self._twin[k] = DiscriminatorNet(self.ops, **self._ctor)
            self._twin[k].wgrad_stream = self.wgrad_stream
        return self._twin[k]

    def set_low(self, low):
        """low [B,T,S,S,cl] -> channels [0:cl] of the mix buffer (constant over a train step)."""
        b = self.buffers(low.shape[0])
        self.to_time_major(low, b["mix"])

    def set_high_tm(self, high_tm, B):
        """high_tm: time-major [T*B,S,S,>=ch] view (e.g. the generator's output buffer)."""
        b = self.buffers(B)
        chp = round4(self.ch)
        if high_tm.shape[-1] == chp and high_tm.stride(-1) == 1:
            # the caller's buffer already has the layout of the high-res branch input (channels padded to 4 with zeros)
            # and stays untouched until this pass' backward has run: read it in place instead of copying it
            b["hi_view"] = high_tm
        else:
            self.ops.copy_channels(high_tm[..., :self.ch], b["hi"][..., :self.ch])
            b["hi_view"] = b["hi"]
        # models.py:100: concat(low, high).  Where the low + high ConvLSTM can read its last `ch` channels from a second tensor
        # (single timestep, fused kernels: ConvLSTM.x2_ok) the high-res part is read in place from hi_view and the concatenation is
        # never completed — its low-res part was written once per step by set_low; otherwise the two channels are copied in
        if self.mix_in_place and self.lstm_b.x2_ok(self.T, self.ch):
            b["mix_x2"] = (b["hi_view"], self.ch)
        else:
            b["mix_x2"] = None
            self.ops.copy_channels(high_tm[..., :self.ch], b["mix"][..., self.cl:self.cl + self.ch])

    def forward(self, B, training, prepared=False):
        """Scores [B] for the resident (low, high) buffers.  prepared=True: the caller has run _prepare(training) for this
        call already (the trainer's two-network critic schedule separates the weight preparation from the pass)."""
        b = self.buffers(B)
        o, Fd, T = self.ops, self.Fd, self.T
        if not prepared:
            self._prepare(training)
        overlap = (self.overlap_branches_tn if T > 1 else self.overlap_branches_t1) and self.overlap_branches
        self._overlap_now = overlap
        if overlap:
            # the two input branches are independent chains of small per-timestep launches: run the high-res-only one
            # on a side stream under the other
            with o.fork() as side:
                self.lstm_a.forward(b["hi_view"], b["ha"], B, T)
                self._conv_ln_fwd(self.conv_a, self.ln_a, b["ha"], b["ya"], b["cat"][..., :Fd])
            self.lstm_b.forward(b["mix"], b["hb"], B, T, x2=b.get("mix_x2"))
            self._conv_ln_fwd(self.conv_b, self.ln_b, b["hb"], b["yb"], b["cat"][..., Fd:])
            side.join()
        elif T > 1 and b.get("mix_x2") is None:
            # n_timesteps > 1: the two recurrences are independent chains of T - 1 dependent steps — one launch per timestep for both
            # (layers.convlstm_pair_forward; falls back to the layers' own loops where the joint step is not available)
            convlstm_pair_forward(self.lstm_a, b["hi_view"], b["ha"], self.lstm_b, b["mix"], b["hb"], B, T)
            self._conv_ln_fwd(self.conv_a, self.ln_a, b["ha"], b["ya"], b["cat"][..., :Fd])
            self._conv_ln_fwd(self.conv_b, self.ln_b, b["hb"], b["yb"], b["cat"][..., Fd:])
        else:
            self.lstm_a.forward(b["hi_view"], b["ha"], B, T)
            self._conv_ln_fwd(self.conv_a, self.ln_a, b["ha"], b["ya"], b["cat"][..., :Fd])
            self.lstm_b.forward(b["mix"], b["hb"], B, T, x2=b.get("mix_x2"))
            self._conv_ln_fwd(self.conv_b, self.ln_b, b["hb"], b["yb"], b["cat"][..., Fd:])
        x = b["cat"]
        for i, (conv, ln, osz, co) in enumerate(self.blocks):
            conv.forward_ln(x, b["ys"][i], b["zs"][i], ln)                            # :113-116 / 122-125 / 134-136
            if self.shortcut is not None and i == self.shortcut["block"]:
                self._shortcut_fwd(b, x, b["zs"][i])
            x = b["zs"][i]
        self._last = x
        o.dense_gap_fwd(x.view(T * B, self.K), self.dense_w.value.view(-1), self.dense_b.value, b["score"], B, T)
        return b["score"]

    # ---- split connection (models.py:118,127-130; tf_utils.py:15-32), shortcut_variant only -------------------
    # The shortcut conv has stride >= kernel (e.g. 6x6 stride 11 from a 9x9 map): its windows are disjoint, so it
    # runs as a 1x1 convolution on the gathered windows with the weights viewed as [k*k*Cin][Cout].
    _G1 = ConvGeom(1, 1, 1, 0)

    def _shortcut_fwd(self, b, x_src, z_main):
        sc, o = self.shortcut, self.ops
        o.patch_gather(x_src, b["sc_patch"], sc["k"], sc["stride"], sc["pad"])
        o.conv_fwd(b["sc_patch"], sc["conv"].pk.as_1x1(), sc["conv"].b.value, b["sc_y"], self._G1, act=True, slope=LRELU)
        sc["ln"].forward(v2(b["sc_y"]), v2(b["sc_z"]))
        o.copy_channels(b["sc_z"], z_main, accumulate=True)                       # kl.add([x, shortcut]), :130

    def _shortcut_bwd(self, b, dsum, x_src, dx_src, need_wgrad):
        """dsum: gradient w.r.t. the sum (a copy: consumed); accumulates the shortcut's share into dx_src."""
        sc, o = self.shortcut, self.ops
        conv, pk1 = sc["conv"], sc["conv"].pk.as_1x1()
        sc["ln"].backward(v2(dsum), v2(b["sc_y"]), v2(dsum), conv.b.grad if need_wgrad else None, need_wgrad)
        if need_wgrad:
            # (the `fresh` protocol of ParamStore.zero_grad(lazy=True), as Conv.backward_weights: a slot left unfilled is
            # stored to by its first writer — this launch is the shortcut kernel's only one)
            acc = not conv.w.fresh
            conv.w.fresh = False
            o.conv_wgrad(b["sc_patch"], dsum, pk1, conv.w.grad.view(1, 1, pk1.cin, pk1.cout), self._G1, accumulate=acc)
        o.conv_dgrad(dsum, pk1, b["sc_dpatch"], self._G1)
        o.patch_scatter(b["sc_dpatch"], dx_src, sc["k"], sc["stride"], sc["pad"], accumulate=True)

    def _fused_conv_ln(self, conv):
        return self.ops.convln_supported(conv.cin, conv.cout)

    def _conv_ln_fwd(self, conv, ln, x, y, z):
        """SN-Conv2D 3x3 + LeakyReLU + LayerNormalization of one branch (models.py:94-97 / 102-105)."""
        if self._fused_conv_ln(conv):
            # z only: the backward recomputes the pre-norm activation and its statistics from the 2-channel input
            self.ops.convln_fwd(x, conv.w.value, conv.b.value, ln.gamma.value, ln.beta.value, LN_EPS, LRELU, None, z, None)
        else:
            conv.forward_ln(x, y, z, ln)     # (z: a channel slice of the concatenation — wdg_conv_fwd_ln_strided)

    def _conv_ln_bwd(self, conv, ln, dz, y, x, dx, need_wgrad, ln_done=False):
        """ln_done: dz has been through the norm's backward already (in the epilogue of the launch that produced it)."""
        if ln_done:
            if need_wgrad:
                self._wgrad(lambda: conv.backward_weights(x, dz), self._bwd_joins)
            conv.backward_input(dz, dx)
            return
        if self._fused_conv_ln(conv):
            if need_wgrad and conv.w.fresh:          # (this kernel accumulates: honour a lazily zeroed slot)
                conv.w.grad.zero_()
                conv.w.fresh = False
            self.ops.convln_bwd_x(dz, x, conv.w.value, conv.b.value, ln.gamma.value, LN_EPS, LRELU, dx,
                                  ln.gamma.grad if need_wgrad else None, ln.beta.grad if need_wgrad else None,
                                  conv.b.grad if need_wgrad else None, conv.w.grad if need_wgrad else None)
        else:
            ln.backward(v2(dz), v2(y), v2(dz), conv.b.grad if need_wgrad else None, need_wgrad)
            if need_wgrad:
                self._wgrad(lambda: conv.backward_weights(x, dz), self._bwd_joins)
            conv.backward_input(dz, dx)

    def backward(self, B, dscore, need_wgrad, need_input_grad=True):
        """dscore [B].  Returns the time-major gradient w.r.t. the high-res input [T*B,S,S,round4(ch)].
        need_wgrad=False is the input-gradient-only pass (gradient penalty and generator step,
        ganbase.py:35,60); need_input_grad=False the weights-only pass of the critic update (ganbase.py:46: the tape
        asks for the discriminator's weights only), which skips the two ConvLSTMs' input gradients and returns None."""
        b = self.buffers(B)
        o, Fd, T = self.ops, self.Fd, self.T
        x = self._last
        N = T * B
        joins = self._bwd_joins = []
        if self.blocks:
            dx = b["dzs"][-1]
        else:
            dx = b["dcat"]
        chain = self.chain_ln_bwd and hasattr(o, "conv_dgrad_lnbwd")
        top_split = self.shortcut is not None and self.shortcut["block"] == len(self.blocks) - 1
        ln_done = False
        if chain and self.blocks and not top_split and self.final_ch % 4 == 0 and self.final_ch <= 1024:
            # the head's backward and the top block's LayerNorm backward in one launch (dz = dscore / T * w stays in registers)
            tconv, tln = self.blocks[-1][0], self.blocks[-1][1]
            o.dense_gap_bwd_ln(x.view(N, self.K), self.dense_w.value.view(-1), dscore, dx.view(N, self.K),
                               self.dense_w.grad.view(-1) if need_wgrad else None, self.dense_b.grad if need_wgrad else None, B, T,
                               b["ys"][-1], tln.mean_rstd, tln.gamma.value, self.final_ch, LRELU,
                               tln.gamma.grad if need_wgrad else None, tln.beta.grad if need_wgrad else None,
                               tconv.b.grad if need_wgrad else None, tln.lnbwd_scratch() if need_wgrad else None)
            ln_done = True
        else:
            o.dense_gap_bwd(x.view(N, self.K), self.dense_w.value.view(-1), dscore, dx.view(N, self.K),
                            self.dense_w.grad.view(-1) if need_wgrad else None,
                            self.dense_b.grad if need_wgrad else None, B, T)
        # The LayerNormalization backward of a block runs in the epilogue of the data gradient that PRODUCES its dz — the
        # data gradient of the block above it (Conv.backward_input_through_ln: the epilogue holds dz for all channels of a
        # pixel, so the norm's two reductions run on the accumulators and the standalone pass over dz disappears) — wherever
        # nothing else adds to that dz afterwards (the shortcut's share arrives by a later accumulate).  ln_done: the dz this
        # iteration starts from has been through its norm's backward already.
        b_chained = False
        for i in range(len(self.blocks) - 1, -1, -1):
            conv, ln, osz, co = self.blocks[i]
            dz = b["dzs"][i]
            split = self.shortcut is not None and i == self.shortcut["block"]
            if split:
                assert not ln_done
                o.copy_channels(dz, b["sc_dz"])                                   # the sum's gradient feeds both branches
            if not ln_done:
                ln.backward(v2(dz), v2(b["ys"][i]), v2(dz), conv.b.grad if need_wgrad else None, need_wgrad)
            xin = b["zs"][i - 1] if i > 0 else b["cat"]
            dxin = b["dzs"][i - 1] if i > 0 else b["dcat"]
            if need_wgrad:
                self._wgrad(lambda conv=conv, xin=xin, dz=dz: conv.backward_weights(xin, dz), joins, small=osz <= 32)
            below_split = self.shortcut is not None and i - 1 == self.shortcut["block"]    # (its dz is copied for the shortcut first)
            ln_done = False
            if chain and not split and not below_split and i > 0:
                pconv, pln = self.blocks[i - 1][0], self.blocks[i - 1][1]
                conv.backward_input_through_ln(dz, dxin, pln, b["ys"][i - 1], 0, pconv.b.grad if need_wgrad else None, need_wgrad)
                ln_done = True
            elif chain and not split and i == 0 and not self._fused_conv_ln(self.conv_b):
                # dcat = [dz of ln_a | dz of ln_b]: the low + high branch's norm (models.py:105) in this launch's epilogue
                conv.backward_input_through_ln(dz, dxin, self.ln_b, b["yb"], Fd, self.conv_b.b.grad if need_wgrad else None, need_wgrad)
                b_chained = True
            else:
                conv.backward_input(dz, dxin)
            if split:
                self._shortcut_bwd(b, b["sc_dz"], xin, dxin, need_wgrad)
        def branch_a():   # high-res only
            self._conv_ln_bwd(self.conv_a, self.ln_a, b["dcat"][..., :Fd], b["ya"], b["ha"], b["dha"], need_wgrad)
            self.lstm_a.backward(b["hi_view"], b["ha"], b["dha"], b["dhi"] if need_input_grad else None, B, T, need_wgrad)

        # d(high) = d(hi) + d(mix)[cl:cl+ch].  Where the low + high layer can produce the gradient of its high-resolution channels
        # alone (one timestep, no weight gradient in this pass: the gradient-penalty pass and the generator step), it adds them
        # onto d(hi) directly — the three low-resolution channels' gradient is never used, and the two channel copies go away
        direct = (need_input_grad and not getattr(self, "_overlap_now", False) and os.environ.get("WDG_DHIGH_DIRECT", "1") != "0"
                  and self.lstm_b.dx_from_ok(T, self.cl, need_wgrad, b.get("mix_x2")))

        def branch_b():   # low + high
            self._conv_ln_bwd(self.conv_b, self.ln_b, b["dcat"][..., Fd:], b["yb"], b["hb"], b["dhb"], need_wgrad, ln_done=b_chained)
            if direct:
                self.lstm_b.backward(b["mix"], b["hb"], b["dhb"], b["dhi"], B, T, need_wgrad, accumulate_dx=True, dx_c0=self.cl)
                return
            self.lstm_b.backward(b["mix"], b["hb"], b["dhb"], b["dmix"] if need_input_grad else None, B, T, need_wgrad,
                                 x2=b.get("mix_x2"))

        if getattr(self, "_overlap_now", False):
            with o.fork() as side:
                branch_a()
            branch_b()
            side.join()
        elif T > 1 and b.get("mix_x2") is None:
            self._conv_ln_bwd(self.conv_a, self.ln_a, b["dcat"][..., :Fd], b["ya"], b["ha"], b["dha"], need_wgrad)
            self._conv_ln_bwd(self.conv_b, self.ln_b, b["dcat"][..., Fd:], b["yb"], b["hb"], b["dhb"], need_wgrad, ln_done=b_chained)
            convlstm_pair_backward(self.lstm_a, b["hi_view"], b["ha"], b["dha"], b["dhi"] if need_input_grad else None,
                                   self.lstm_b, b["mix"], b["hb"], b["dhb"], b["dmix"] if need_input_grad else None, B, T, need_wgrad)
        else:
            branch_a()
            branch_b()
        self._join(joins)
        if not need_input_grad:
            return None
        if direct:
            return b["dhi"]
        # d(high) = d(hi) + d(mix)[cl:cl+ch]
        o.copy_channels(b["dhi"][..., :self.ch], b["dhigh"][..., :self.ch])
        o.copy_channels(b["dmix"][..., self.cl:self.cl + self.ch], b["dhigh"][..., :self.ch], accumulate=True)
        return b["dhigh"]


class EncoderNet(_Net):
    """AutoEncoder.make_encoder (/root/reference/src/downscaling/autoencoder/autoencoder.py:23-36): the feature
    extractor of `reconstruction_loss` (gan/train.py:19-26, ganbase.py:57-59).  While the map is >= 7 pixels:
    ZeroPadding2D(1) -> SN Conv2D(2C, 5x5, stride 3) -> LeakyReLU(0.2) -> LayerNormalization; Flatten; Dense((flat +
    latent) // 2) if flat > 2 * latent; Dense(latent).  Forward and the input gradient (the generator step
    differentiates the loss w.r.t. the generated winds only; the extractor's weights are frozen there)."""

    def __init__(self, ops, img_size, n_timesteps, latent_dimension, in_channels=2, seed=3):
        super().__init__(ops, None)
        self.S, self.T, self.latent, self.cin = img_size, n_timesteps, latent_dimension, in_channels
        L = "layer_with_weights-"
        size, ch, idx = img_size, in_channels, 0
        self.blocks = []
        while size >= 7:                                                               # autoencoder.py:26
            osz = (size + 2 - 5) // 3 + 1
            conv = self._add(Conv(self, L + str(idx), 5, ch, ch * 2, 3, 1, sn=True))   # :27-29
            ln = self._add(LayerNorm(self, L + str(idx + 1), ch * 2))                  # :30
            self.blocks.append((conv, ln, osz, ch * 2))
            size, ch, idx = osz, ch * 2, idx + 2
        if ch % 4 != 0:
            raise NotImplementedError("encoder: the flattened map needs a channel count that is a multiple of 4")
        self.flat = size * size * ch
        self.final = (size, ch)
        self.dense = []
        K = self.flat
        if K > 2 * latent_dimension:                                                   # :32-34
            mid = (K + latent_dimension) // 2
            self.dense.append(self._add(Dense(self, L + str(idx), K, mid)))
            K, idx = mid, idx + 1
        self.dense.append(self._add(Dense(self, L + str(idx), K, latent_dimension)))   # :35
        self._finalize(seed)

    def buffers(self, B):
        b = self._bufs.get(B)
        if b is not None:
            return b
        o, N = self.ops, self.T * B
        b = dict(x0=o.zeros(N, self.S, self.S, round4(self.cin)), dx0=o.zeros(N, self.S, self.S, round4(self.cin)),
                 ys=[], zs=[], dzs=[], hs=[], dhs=[])
        for (_, _, osz, co) in self.blocks:
            b["ys"].append(o.empty(N, osz, osz, co))
            b["zs"].append(o.empty(N, osz, osz, co))
            b["dzs"].append(o.empty(N, osz, osz, co))
        for d in self.dense:
            b["hs"].append(o.zeros(N, 1, 1, round4(d.units)))
            b["dhs"].append(o.zeros(N, 1, 1, round4(d.units)))
        self._bufs = {B: b}
        return b

    def forward(self, x):
        """x [B,T,S,S,cin] -> latent [B,T,latent] (inference mode: spectral normalisation inactive)."""
        B = x.shape[0]
        b, N = self.buffers(B), self.T * x.shape[0]
        self._prepare(False)
        self.to_time_major(x, b["x0"])
        h = b["x0"]
        for i, (conv, ln, osz, co) in enumerate(self.blocks):
            conv.forward_ln(h, b["ys"][i], b["zs"][i], ln)
            h = b["zs"][i]
        h = h.view(N, 1, 1, self.flat) if self.blocks else h.reshape(N, 1, 1, -1)
        for j, d in enumerate(self.dense):
            d.forward(h, b["hs"][j])
            h = b["hs"][j]
        out = self.ops.empty(B, self.T, 1, 1, self.latent)
        self.from_time_major(h, out)
        return out.view(B, self.T, self.latent)

    def backward_input(self, dlatent):
        """dlatent [B,T,latent] for the activations of the LAST forward -> d/dx [B,T,S,S,cin]."""
        B = dlatent.shape[0]
        b, N, o = self.buffers(B), self.T * dlatent.shape[0], self.ops
        dh = b["dhs"][-1]
        self.to_time_major(dlatent.reshape(B, self.T, 1, 1, self.latent), dh)
        for j in range(len(self.dense) - 1, -1, -1):
            if j > 0:
                dst = b["dhs"][j - 1]
            else:
                dst = b["dzs"][-1].view(N, 1, 1, self.flat) if self.blocks else b["dx0"].view(N, 1, 1, -1)
            self.dense[j].backward_input(dh, dst)
            dh = dst
        for i in range(len(self.blocks) - 1, -1, -1):
            conv, ln, osz, co = self.blocks[i]
            dz = b["dzs"][i]
            ln.backward(v2(dz), v2(b["ys"][i]), v2(dz), None, False)
            conv.backward_input(dz, b["dzs"][i - 1] if i > 0 else b["dx0"])
        dx = o.empty(B, self.T, self.S, self.S, self.cin)
        self.from_time_major(b["dx0"], dx)
        return dx
