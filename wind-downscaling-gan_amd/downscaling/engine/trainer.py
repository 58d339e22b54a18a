"""GAN train / test step of the reference (/root/reference/src/downscaling/gan/ganbase.py:21-113) on
the layer engine, plus the TF-form Adam optimizer (gan/train.py:34-35,57-58), the Philox noise
source behind FlexibleNoiseGenerator (data/data_generator.py:319-335) and the data-parallel glue.
"""
import math
import os

import torch

from .common import round4, v2


class AdamTF:
    """tf.keras.optimizers.Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps)."""

    def __init__(self, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, learning_rate=None):
        self.lr = float(learning_rate if learning_rate is not None else lr)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        self.iterations = 0
        self.m = self.v = None

    def apply_gradients(self, store, grad_scale=1.0):
        ops = store.ops
        if self.m is None:
            self.m = ops.zeros(store.flat.numel())
            self.v = ops.zeros(store.flat.numel())
        self.iterations += 1
        t = self.iterations
        lr_t = self.lr * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)
        store.settle()
        ops.adam_tf(store.flat, store.grads, self.m, self.v, lr_t, self.beta_1, self.beta_2, self.epsilon, grad_scale)
        store.version += 1


class PhiloxSource:
    """Counter-based normal/uniform source: (seed, running counter offset).  Every draw advances the
    offset by the number of Philox blocks it consumed, so a sequence of draws is reproducible."""

    def __init__(self, ops, seed=None, rank=0):
        self.ops = ops
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
        self.seed = (int(seed) + 0x9E3779B97F4A7C15 * int(rank)) & (2 ** 64 - 1)
        self.offset = 0

    def reserve(self, n_elements):
        """Claims the next ceil(n / 4) Philox blocks of the stream and returns their offset: a draw can then be LAUNCHED later
        (or on another HIP stream) than its place in the reference's draw order."""
        off = self.offset
        self.offset += (int(n_elements) + 3) // 4
        return off

    def normal_at(self, view2d, std, offset, add=None):
        self.ops.philox_normal(view2d, self.seed, offset, float(std), add)

    def assemble_at(self, image, rows_out, B, XY, cn, std, offset):
        """[image | noise | 0] into whole rows of the generator's input buffer; the noise is normal_at's stream at `offset`."""
        self.ops.input_assemble(image, rows_out, B, XY, cn, self.seed, offset, float(std))

    def assemble_slots_at(self, image, rows_all, B, XY, cn, std, offset, Bo, b0):
        """assemble_at for batch slots [b0, b0 + B) of a time-major buffer of Bo slots (HipOps.input_assemble_slots); backends
        without it get one assemble_at per timestep (each timestep's rows of the group are one contiguous block)."""
        fn = getattr(self.ops, "input_assemble_slots", None)
        if fn is not None:
            fn(image, rows_all, B, XY, cn, self.seed, offset, float(std), Bo, b0)
            return
        rows = B * XY
        for t in range(image.shape[1]):
            r0 = (t * Bo + b0) * XY
            self.assemble_at(image[:, t:t + 1], rows_all[r0:r0 + rows], B, XY, cn, std, offset + t * (rows * cn // 4))

    def uniform_at(self, vec, offset):
        self.ops.philox_uniform(vec, self.seed, offset)

    def normal_into(self, view2d, std, add=None):
        self.normal_at(view2d, std, self.reserve(view2d.shape[0] * view2d.shape[1]), add)

    def uniform_into(self, vec):
        self.uniform_at(vec, self.reserve(vec.numel()))


class DistSync:
    """One process per GPU; RCCL (backend "nccl") all-reduce over xGMI.  The path has exactly two
    exchange points: the flat gradient buffers (once per optimizer step) and the BatchNorm batch
    statistics (SyncBN, so a global batch of 8x32 normalises like the single-device reference)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # WDG_DIST_ALWAYS=1: run the exchange code path (async all-reduce, deferred Adam, SyncBN, metric reduce) even
        # in a one-rank group — a single MI355X then executes the real RCCL collectives (identity results), which is how
        # the `-m gpu` suite covers the "nccl" branch on a one-GPU box
        self.active = self.world_size > 1 or os.environ.get("WDG_DIST_ALWAYS", "0") == "1"

        # gloo has no device collectives in every build: stage device tensors through the host there (tests that
        # run two ranks on one GPU; the production backend is "nccl" = RCCL, which reduces in place on the device)
        self._stage = self.dist.get_backend(group) == "gloo"
        # measurement switch (bench A/B only): keep the whole exchange code path (deferred Adam, SyncBN bookkeeping, metric
        # reduce) but issue no collective — separates the cost of the schedule from the cost of the RCCL calls
        self._noop = os.environ.get("WDG_DIST_NOOP", "0") == "1"
        if self._noop:
            if self.world_size > 1 and os.environ.get("WDG_DIST_NOOP_FORCE", "0") != "1":
                raise RuntimeError("WDG_DIST_NOOP=1 in a multi-rank job: every all-reduce (gradients, SyncBN statistics, metrics) "
                                   "would be skipped and the replicas would train on rank-local gradients.  It is a one-rank "
                                   "measurement switch; unset it (or set WDG_DIST_NOOP_FORCE=1 for a timing-only run).")
            import warnings
            warnings.warn("WDG_DIST_NOOP=1: collectives are skipped (timing experiment only, results are not a training step)",
                          RuntimeWarning)

        # measurement switch (bench.py dp_fifth_queue_proxy, one rank only): every collective is replaced by a stand-in kernel
        # that occupies the chip the way the 8-rank collective would (wdg_dp_proxy) — the large gradient all-reduces as 16
        # workgroups on a stream of their own (the FIFTH busy queue beside main / generator / real / generated) moving
        # 2 * 7/8 of the buffer and lasting what the ring needs at ~150 GB/s per xGMI link; the small blocking ones (SyncBN
        # statistics, metrics) as a ~10 us wait on the calling stream.  Results are those of a one-rank group (identity).
        self._proxy = os.environ.get("WDG_DP_PROXY", "0") == "1"
        if self._proxy:
            if self.world_size > 1:
                raise RuntimeError("WDG_DP_PROXY=1 is a one-rank measurement switch (it replaces the collectives by stand-in kernels)")
            self.active = True
            self._proxy_stream = self._proxy_scratch = None
            self.proxy_ranks = int(os.environ.get("WDG_DP_PROXY_RANKS", "8"))
            self.proxy_link_gbps = float(os.environ.get("WDG_DP_PROXY_GBPS", "150"))
            self.proxy_small_us = float(os.environ.get("WDG_DP_PROXY_SMALL_US", "10"))
            self.proxy_blocks = int(os.environ.get("WDG_DP_PROXY_BLOCKS", "16"))
            self.proxy_log = {"large": 0, "small": 0, "large_bytes": 0}

    def _proxy_large(self, t):
        from downscaling.engine import native
        lib = native.load()
        cur = torch.cuda.current_stream(t.device)
        if self._proxy_stream is None:
            self._proxy_stream = torch.cuda.Stream(device=t.device)
        nb = t.numel() * t.element_size()
        if self._proxy_scratch is None or self._proxy_scratch.numel() < nb:
            self._proxy_scratch = torch.empty(nb, dtype=torch.uint8, device=t.device)
        n = self.proxy_ranks
        moved = int(2 * (n - 1) / n * nb) // 16 * 16
        s = self._proxy_stream
        s.wait_stream(cur)
        native.check(lib.wdg_dp_proxy(t.data_ptr(), self._proxy_scratch.data_ptr(), nb // 16 * 16, moved, self.proxy_blocks,
                                      moved / (self.proxy_link_gbps * 1e3), s.cuda_stream), "dp_proxy")
        ev = torch.cuda.Event()
        ev.record(s)
        self.proxy_log["large"] += 1
        self.proxy_log["large_bytes"] += moved
        return lambda: torch.cuda.current_stream(t.device).wait_event(ev)

    def _proxy_small(self, t):
        from downscaling.engine import native
        native.check(native.load().wdg_dp_proxy(None, None, 0, 0, 1, self.proxy_small_us, torch.cuda.current_stream(t.device).cuda_stream), "dp_proxy")
        self.proxy_log["small"] += 1

    def all_reduce_sum_async(self, t):
        """Start the all-reduce and return a zero-argument `finish()`; the collective runs on RCCL's own stream, so
        kernels enqueued on the compute stream before `finish()` overlap with it (`finish` makes the compute stream
        wait for the result)."""
        if self._noop:
            return lambda: None
        if self._proxy and t.is_cuda:
            if os.environ.get("WDG_DP_PROXY_LARGE", "1") == "0":       # (A/B: only the small blocking stand-ins)
                return lambda: None
            return self._proxy_large(t)
        if self._stage and t.is_cuda:
            self.all_reduce_sum(t)          # host-staged test path: nothing to overlap
            return lambda: None
        work = self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        return work.wait

    def all_reduce_sum(self, t):
        if self._noop:
            return
        if self._proxy:
            if t.is_cuda:
                self._proxy_small(t)
            return
        if self._stage and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)


class GanEngine:
    GAMMA = 100.0  # ganbase.py:22

    def __init__(self, gen, disc, noise_source, noise_std, n_critic=3, sync=None, sync_bn=True):
        self.gen, self.disc, self.noise, self.noise_std = gen, disc, noise_source, float(noise_std)
        self.n_critic = n_critic
        self.sync = sync
        self.ops = gen.ops
        if sync is not None and sync_bn:
            gen.sync = sync
        self._tmp = {}
        self._pending = {}
        # the generator forward of critic iteration i + 1 (and of the generator step) runs on its own HIP stream under the
        # discriminator passes of iteration i, D(real) of the metrics recompute under the generator's backward
        # (WDG_OVERLAP_GEN=0 disables) — _critic_pipelined.  The data-parallel step runs the SAME schedule: the gradient
        # all-reduce of iteration i is started where the single-process step runs Adam and lands at the next _flush(disc), the
        # generator stream runs under it; SyncBN's small all-reduces are issued from the generator's stream
        self.overlap_generator = os.environ.get("WDG_OVERLAP_GEN", "1") != "0"
        self._gen_stream = None
        # the discriminator's gradient-penalty pass beside its real pass on a twin network (WDG_OVERLAP_DISC=0 disables):
        # 69.0 -> 68.6 ms at the headline shape, +0.8 % at T = 24 (same-box A/B) — see _critic_pipelined
        self.overlap_discriminator = os.environ.get("WDG_OVERLAP_DISC", "1") != "0"
        # WDG_OVERLAP_DISC=2 (default): the generated pass on a THIRD network and stream as well — all three passes of an
        # iteration side by side, main / generator / real / generated = the four hardware queues; the weight gradients of the two
        # copies then stay on their pass' stream and the generator's go to the (then idle) real-pass stream: a fifth stream costs
        # more than it hides.  Same box, alternating: 62.9 -> 60.8 ms at the headline shape, 80.4 -> 76.7 ms at T = 24
        # (profiles/r05aa_ab_step_triple.txt ... r05ac); =1: two networks (rounds 4-5)
        self.triple_discriminator = os.environ.get("WDG_OVERLAP_DISC", "2")
        self._fake_stream = None
        only = os.environ.get("WDG_WGRAD_STREAM_ONLY")           # A/B switch: "g" / "d" keep the weight-gradient stream on one network
        if only == "g":
            disc.wgrad_stream = False
        elif only == "d":
            gen.wgrad_stream = False
        self.hoist_first_real_pass = os.environ.get("WDG_HOIST_REAL", "0") != "0"      # measured: 66.45 vs 66.25 ms with it (profiles/r04s_hoist.txt) - big kernels of two networks side by side gain nothing
        self._disc_stream = None
        self.assemble_input = os.environ.get("WDG_ASSEMBLE_INPUT", "0") != "0"     # (A/B switch of _gen_noise_at: measured neutral in the train step - 64.6 vs 64.4 ms, the draw runs on the generator stream under the discriminator passes - and left off)

    def _buf(self, key, *shape):
        t = self._tmp.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self.ops.zeros(*shape)
            self._tmp[key] = t
        return t

    def _const(self, key, n, value):
        """A length-n vector holding `value` (the score gradients +-sw/B, the ones of the gradient-penalty pass): filled when
        the value changes, not once per use — seven torch fill launches per step less."""
        ent = self._tmp.get(("const", key))
        if ent is None or ent[0].shape[0] != n or ent[1] != value:
            t = ent[0] if ent is not None and ent[0].shape[0] == n else self.ops.zeros(n)
            t.fill_(value)
            ent = self._tmp[("const", key)] = (t, value)
        return ent[0]

    def _gen_noise_at(self, low, B, offset):
        """The generator's noise channels (ganbase.py:28,51,65) drawn at Philox offset `offset`.  Where the backend has the
        one-pass input assembly (wdg_input_assemble: [image | noise | 0] per pixel with 16-byte stores, the same counters
        and arithmetic as philox_normal) the whole input row is rewritten — 104 -> ~60 us per draw at the headline shape:
        the strided 4-byte stores of the noise-only kernel are what it is bound by; otherwise the noise slice alone."""
        gen, noise = self.gen, self.noise
        ok = getattr(self.ops, "input_assemble_ok", None)
        rows = gen.input_rows(B)
        if self.assemble_input and ok is not None and low.shape[-1] == gen.in_channels and ok(gen.in_channels, gen.noise_channels, rows.shape[1]):
            noise.assemble_at(low, rows, B, gen.S * gen.S, gen.noise_channels, self.noise_std, offset)
        else:
            noise.normal_at(gen.noise_view(B), self.noise_std, offset)

    def _gen_noise(self, low, B):
        nv = self.gen.noise_view(B)
        self._gen_noise_at(low, B, self.noise.reserve(nv.shape[0] * nv.shape[1]))

    def _reduce_and_step(self, net, opt):
        """Gradient all-reduce + optimizer step.  With several ranks the all-reduce is started asynchronously and the
        optimizer step is deferred until the network is next touched (`_flush`): the generator forward that opens the
        next critic iteration does not depend on the discriminator's weights (and the discriminator pass that opens the
        metrics recompute not on the generator's), so that compute hides the exchange."""
        scale = 1.0
        net.params.settle()
        if self.sync is not None and self.sync.active:
            scale = 1.0 / self.sync.world_size
            finish = self.sync.all_reduce_sum_async(net.params.grads)
            self._pending[id(net)] = (net, opt, scale, finish)
            return scale
        opt.apply_gradients(net.params, grad_scale=scale)
        return scale

    def _flush(self, net):
        """Complete a deferred exchange + optimizer step of `net` (no-op when nothing is pending)."""
        pend = self._pending.pop(id(net), None)
        if pend is not None:
            _, opt, scale, finish = pend
            finish()
            opt.apply_gradients(net.params, grad_scale=scale)

    # keys whose value is a mean over THIS rank's batch shard; the gradient-norm metrics are formed from the all-reduced
    # gradients and are global already
    _SHARD_MEANS = ("g_loss", "g_disc_loss", "g_reco_loss", "d_loss", "d_gradient_pen", "_d_loss_train", "_d_real",
                    "_d_fake", "loss")

    def _reduce_metrics(self, res):
        """ganbase.py:75-81 logs losses of the WHOLE batch.  Under batch data-parallelism every rank holds the mean over
        its equal-sized shard, so the global value is the mean over ranks: one small all-reduce of the stacked scalars
        (SURVEY 8e), after which every rank returns identical logs."""
        if self.sync is None or not self.sync.active:
            return res
        keys = [k for k in self._SHARD_MEANS if res.get(k) is not None]
        vec = torch.stack([res[k].reshape(()) for k in keys])
        self.sync.all_reduce_sum(vec)
        vec /= self.sync.world_size
        for i, k in enumerate(keys):
            res[k] = vec[i]
        return res

    def _grad_param_metric(self, net, scale):
        st = net.params
        self.ops.segment_meansq(st.grads, st.seg_pairs, st.seg_out)
        per_var = st.seg_out if st.seg_scale is None else st.seg_out * st.seg_scale.to(st.seg_out.dtype)
        return per_var.mean() * (scale * scale)

    @staticmethod
    def _compiled_loss(value, sample_weight, sw_mean):
        """What Keras' `compiled_loss(y_true, y_pred, sample_weight)` makes of a plain-function loss (compute_weighted_loss,
        SUM_OVER_BATCH_SIZE): a scalar value times the mean weight; a per-sample vector weighted ELEMENTWISE, summed, divided by its
        element count (oracle/torch_model.py::weighted_loss)."""
        if value.numel() <= 1:
            return value.reshape(()) * sw_mean
        v = value.reshape(-1)
        if sample_weight is not None:
            v = v * torch.as_tensor(sample_weight, dtype=v.dtype, device=v.device).reshape(-1)
        return v.sum() / v.numel()

    def _critic_coupled(self, d_loss_fn, B, real, fake, noisy, sw_mean, low, sample_weight=None):
        """Real + generated pass of one critic iteration for an arbitrary compiled loss d_loss_fn(real_output,
        fake_output) (ganbase.py:41-46).  The built-in Wasserstein loss is separable, so its real pass is differentiated
        before the generated pass runs; a general loss needs both score vectors first.  The real pass therefore runs on
        the discriminator's twin (same variable values: copied, then the same deterministic SN update), which keeps that
        pass' weights and activations while this network takes the SN update of the generated pass; the loss is
        differentiated w.r.t. the two score vectors by torch autograd (2 x B numbers) and each pass is back-propagated
        with its own scores' gradient.  Returns (loss value, real mean, fake mean)."""
        disc, ops, noise = self.disc, self.ops, self.noise
        ch = disc.ch
        twin = disc.twin()
        twin.params.copy_from(disc.params)
        twin.params.zero_grad()
        twin.set_low(low)
        noise.normal_into(v2(noisy[..., :ch]), self.noise_std, add=v2(real[..., :ch]))        # :40
        twin.set_high_tm(noisy, B)
        real_scores = twin.forward(B, training=True).clone()                                  # :41  (SN update on the copy)
        disc._prepare(True)                                                                   # ... and the same one here
        # (the twin reads `noisy` in place until its backward has run: the generated pass gets its own buffer)
        noisy2 = self._buf("noisy2", *noisy.shape)
        noise.normal_into(v2(noisy2[..., :ch]), self.noise_std, add=v2(fake[..., :ch]))       # :42
        disc.set_high_tm(noisy2, B)
        fake_scores = disc.forward(B, training=True).clone()                                  # :43
        r = real_scores.detach().clone().requires_grad_(True)
        f = fake_scores.detach().clone().requires_grad_(True)
        loss = self._compiled_loss(d_loss_fn(r.view(B, 1), f.view(B, 1)), sample_weight, sw_mean)   # :44 compiled_loss
        gr, gf = torch.autograd.grad(loss, (r, f), allow_unused=True)
        gr = torch.zeros_like(r) if gr is None else gr
        gf = torch.zeros_like(f) if gf is None else gf
        twin.backward(B, gr.contiguous(), need_wgrad=True, need_input_grad=False)
        disc.backward(B, gf.contiguous(), need_wgrad=True, need_input_grad=False)
        disc.params.grads.add_(twin.params.grads)
        return loss.detach(), real_scores.mean(), fake_scores.mean()

    def _critic_pipelined(self, low, B, T, real, comb, noisy, eps, gsq, ones, sw_mean, d_opt):
        """The critic iterations (ganbase.py:26-47) with the generator forward of iteration i + 1 on a second HIP stream under
        the three discriminator passes of iteration i.  Within a train step the generator's weights do not depend on the
        discriminator's, so the only ordering the reference imposes between them is the data: fake_i feeds the interpolate and
        the generated pass of iteration i.  Both consumers run first (the instance-noised copy of fake_i into its own
        buffer), then the generator is free to overwrite its activations.  At the shipped sequence length the discriminator's
        passes are chains of small per-timestep launches that leave most of the chip idle; the generator's large kernels
        fill it.  Every random draw keeps the Philox offset of its place in the reference's draw order (reserved up front:
        generator noise, eps, instance noise of the real pass, of the generated pass, per iteration), so the arithmetic —
        and the oracle replay of tests/helpers.Draws — is unchanged.  Returns (disc_loss, gnorm, dscale, the generator step's
        forward output)."""
        gen, disc, ops, noise = self.gen, self.disc, self.ops, self.noise
        S, ch = gen.S, disc.ch
        N, ppi = T * B, S * S
        nview = gen.noise_view(B)
        n_g, n_i = nview.shape[0] * nview.shape[1], N * ppi * ch
        offs = [(noise.reserve(n_g), noise.reserve(B), noise.reserve(n_i), noise.reserve(n_i)) for _ in range(self.n_critic)]
        o_gstep = noise.reserve(n_g)                                              # the generator step's noise (:51) comes next
        nf = self._buf("noisy_fake", *noisy.shape)
        if self._gen_stream is None or self._disc_stream is None:
            # (streams chosen by a measured concurrency probe: on distinct hardware queues whatever else holds pool streams)
            # (the weight-gradient side stream of the networks stays the pool's "wgrad" stream: a third probe-chosen stream measured
            # 0.15 ms worse, 63.44 against 63.28 ms, profiles/r05y_ab_step.txt)
            self._gen_stream, self._disc_stream = ops.concurrent_streams(2)
        gs, main = self._gen_stream, torch.cuda.current_stream(ops.device)
        twin = disc.twin() if self.overlap_discriminator else None
        ds = self._disc_stream
        triple = twin is not None and self.triple_discriminator == "2"
        if triple and self._fake_stream is None:
            self._fake_stream = ops.concurrent_streams(3)[2]
        fs, twin2 = self._fake_stream, (disc.twin(1) if triple else None)
        if triple:
            tw = os.environ.get("WDG_TRIPLE_WGRAD", "0")       # (A/B switch: 1 = the pool's "wgrad" stream, gs / gs_real = the generator's stream)
            twin.wgrad_stream = tw in ("1", "gs", "gs_real")
            twin2.wgrad_stream = tw in ("1", "gs")
            twin.wgrad_side = twin2.wgrad_side = gs if tw.startswith("gs") else None
            if os.environ.get("WDG_WGRAD_STREAM_G", "1") == "1":
                gen.wgrad_side = ds               # the generator's backward runs when the real-pass stream has nothing to do
        else:
            # (two-network / one-network schedules: the generator keeps the pool's side stream, and a twin does not fork its weight
            # gradients onto the stream this network's already use — each fork / join would make one network wait for the other's)
            gen.wgrad_side = None
            if twin is not None:
                twin.wgrad_stream = False

        def start_real_pass(o_r):
            """The three discriminator passes of an iteration on TWO networks (this one and its twin: own variables and
            activations) and two streams.  What orders the passes in the reference is the spectral-norm chain of the weights —
            pass k + 1 reads SN(weights of pass k) — not the passes' results: the real pass' weights are ready as soon as the
            gradient-penalty pass' weights exist.  So: prepare W1 here; the twin takes a copy, prepares W2 = SN(W1) and runs
            the real pass (forward + weights-only backward) on its own stream while the gradient-penalty pass (forward + input
            gradient, W1) runs on the main stream; then this network takes W2 from the twin, prepares W3 and runs the generated
            pass beside the tail of the real one.  The weight gradients of the two passes are summed before the optimizer
            step (R + F, the same sum the single-network form accumulates)."""
            self._flush(disc)
            disc._prepare(True)                                                   # W0 -> W1 (SN of the gradient-penalty pass)
            ds.wait_stream(main)
            with torch.cuda.stream(ds):
                twin.params.copy_from(disc.params)
                twin.params.zero_grad(lazy=True)
                twin._prepare(True)                                               # W1 -> W2 (SN of the real pass)
                w2_ready = torch.cuda.Event()
                w2_ready.record(ds)
                noise.normal_at(v2(noisy[..., :ch]), self.noise_std, o_r, add=v2(real[..., :ch]))   # :40
                twin.set_high_tm(noisy, B)
                real_mean = twin.forward(B, training=True, prepared=True).mean()                    # :41
                real_mean.record_stream(main)                                     # (allocated on `ds`, consumed on the main stream)
                twin.backward(B, self._const("real", B, -sw_mean / B), need_wgrad=True, need_input_grad=False)
            return w2_ready, real_mean

        # The real pass of iteration 0 needs neither fake_0 nor anything the generator writes: it starts BEFORE the first
        # generator forward and runs under it (the only stretch of the critic loop that had one stream busy)
        hoisted = None
        if twin is not None and self.hoist_first_real_pass:
            hoisted = start_real_pass(offs[0][2])
        self._gen_noise_at(low, B, offs[0][0])                                    # :28
        fake = gen.forward(B, training=True, need_backward=False)                 # :29
        for i in range(self.n_critic):
            o_g, o_e, o_r, o_f = offs[i]
            noise.uniform_at(eps, o_e)                                            # :30
            ops.lerp_batch(v2(real), v2(fake), eps, v2(comb), ppi, B)             # :31
            noise.normal_at(v2(nf[..., :ch]), self.noise_std, o_f, add=v2(fake[..., :ch]))      # :42 (drawn now, used below)
            gs.wait_stream(main)                                                  # fake_i has been consumed
            with torch.cuda.stream(gs):
                if i + 1 < self.n_critic:
                    self._gen_noise_at(low, B, offs[i + 1][0])
                    fake = gen.forward(B, training=True, need_backward=False)
                else:
                    # the forward of the GENERATOR step (:50-52) does not read the discriminator either: under the last
                    # iteration's passes
                    gen.params.zero_grad()
                    self._gen_noise_at(low, B, o_gstep)
                    fake = gen.forward(B, training=True, need_backward=True)
            if triple:
                # three networks, three streams: gradient-penalty pass here (W1), real pass on the twin (W2 = SN(W1)), generated
                # pass on the third network (W3 = SN(W2), prepared from the twin's copy as soon as W2 exists).  This network
                # takes W3 and the sum of the two weight-gradient buffers before the optimizer step.
                if i == 0 and hoisted is not None:
                    w2_ready, real_mean = hoisted
                else:
                    w2_ready, real_mean = start_real_pass(o_r)
                fs.wait_stream(main)                                              # nf drawn, the previous iteration's step done
                fs.wait_event(w2_ready)
                with torch.cuda.stream(fs):
                    twin2.params.copy_from(twin.params)                           # W2 (+ its SN state)
                    twin2.params.zero_grad(lazy=True)
                    twin2.set_high_tm(nf, B)
                    fake_mean = twin2.forward(B, training=True).mean()            # :43 (prepares W3 = SN(W2))
                    fake_mean.record_stream(main)
                    twin2.backward(B, self._const("fake", B, sw_mean / B), need_wgrad=True, need_input_grad=False)
                disc.set_high_tm(comb, B)
                disc.forward(B, training=True, prepared=True)                     # :32-34 (W1)
                dcomb = disc.backward(B, ones, need_wgrad=False)                  # :35
                ops.sumsq_batch_ch(v2(dcomb), ppi, T, B, gsq)                     # :36
                gnorm = torch.sqrt(gsq[:, :ch])
                gradient_reg = self.GAMMA * ((gnorm - 1.0) ** 2).mean()           # :37
                main.wait_stream(ds)
                main.wait_stream(fs)
                twin.params.settle()
                twin2.params.settle()
                disc.params.set_grads_sum(twin.params, twin2.params)              # R + F
                disc.params.copy_from(twin2.params)                               # W3
            elif twin is not None:
                if i == 0 and hoisted is not None:
                    w2_ready, real_mean = hoisted
                else:
                    w2_ready, real_mean = start_real_pass(o_r)
                disc.set_high_tm(comb, B)
                disc.forward(B, training=True, prepared=True)                     # :32-34 (W1)
                dcomb = disc.backward(B, ones, need_wgrad=False)                  # :35
                ops.sumsq_batch_ch(v2(dcomb), ppi, T, B, gsq)                     # :36
                gnorm = torch.sqrt(gsq[:, :ch])
                gradient_reg = self.GAMMA * ((gnorm - 1.0) ** 2).mean()           # :37
                disc.params.zero_grad(lazy=True)
                main.wait_event(w2_ready)
                disc.params.copy_from(twin.params)                                # W2 (the twin only reads it from here on)
                disc.set_high_tm(nf, B)
                fake_mean = disc.forward(B, training=True).mean()                 # :43 (prepares W3 = SN(W2))
                disc.backward(B, self._const("fake", B, sw_mean / B), need_wgrad=True, need_input_grad=False)
                main.wait_stream(ds)
                disc.params.settle()
                twin.params.settle()
                disc.params.grads.add_(twin.params.grads)
            else:
                self._flush(disc)
                disc.set_high_tm(comb, B)
                disc.forward(B, training=True)                                    # :32-34
                dcomb = disc.backward(B, ones, need_wgrad=False)                  # :35
                ops.sumsq_batch_ch(v2(dcomb), ppi, T, B, gsq)                     # :36
                gnorm = torch.sqrt(gsq[:, :ch])
                gradient_reg = self.GAMMA * ((gnorm - 1.0) ** 2).mean()           # :37
                disc.params.zero_grad()
                noise.normal_at(v2(noisy[..., :ch]), self.noise_std, o_r, add=v2(real[..., :ch]))   # :40
                disc.set_high_tm(noisy, B)
                real_mean = disc.forward(B, training=True).mean()                 # :41
                disc.backward(B, self._const("real", B, -sw_mean / B), need_wgrad=True, need_input_grad=False)
                disc.set_high_tm(nf, B)
                fake_mean = disc.forward(B, training=True).mean()                 # :43
                disc.backward(B, self._const("fake", B, sw_mean / B), need_wgrad=True, need_input_grad=False)
            disc_loss = (fake_mean - real_mean) * sw_mean + gradient_reg          # :44-45
            dscale = self._reduce_and_step(disc, d_opt)                           # :46-47
            main.wait_stream(gs)
        return disc_loss, gnorm, dscale, fake

    def train_step(self, low, high, g_opt, d_opt, sample_weight=None, reconstruction_loss=None, d_loss_fn=None):
        """One GAN.train_step (ganbase.py:21-94).  low [B,T,S,S,cl], high [B,T,S,S,ch] device tensors.
        d_loss_fn: the discriminator's compiled loss when it is not the built-in Wasserstein form (None = built-in)."""
        gen, disc, ops, noise = self.gen, self.disc, self.ops, self.noise
        B, T = low.shape[0], low.shape[1]
        S, ch = gen.S, disc.ch
        chp = round4(ch)
        N, ppi = T * B, S * S
        sw_mean = 1.0 if sample_weight is None else float(torch.as_tensor(sample_weight).double().mean())
        # (supports_streams: the backend launches on torch's current HIP stream, so work can be forked onto side streams; the
        # oracle backend runs the serial program.  n_critic < 1: nothing to pipeline)
        pipelined = (d_loss_fn is None and getattr(ops, "supports_streams", False) and bool(self.overlap_generator)
                     and self.n_critic >= 1)
        gen.set_image(low)
        disc.set_low(low)
        if pipelined and self.overlap_discriminator:
            disc.twin().set_low(low)          # (the twin network of the two-stream critic schedule reads the same low-res input)
            if self.triple_discriminator == "2":
                disc.twin(1).set_low(low)
        real = self._buf("real", N, S, S, chp)
        gen.to_time_major(high, real)
        comb, noisy = self._buf("comb", N, S, S, chp), self._buf("noisy", N, S, S, chp)
        eps, gsq = self._buf("eps", B), self._buf("gsq", B, chp)
        ones = self._const("ones", B, 1.0)

        if pipelined:
            disc_loss, gnorm, dscale, fake = self._critic_pipelined(low, B, T, real, comb, noisy, eps, gsq, ones, sw_mean, d_opt)
        for _ in range(0 if pipelined else self.n_critic):                        # ganbase.py:26
            self._gen_noise(low, B)                                               # :28
            fake = gen.forward(B, training=True, need_backward=False)             # :29 (outside any tape)
            noise.uniform_into(eps)                                               # :30
            ops.lerp_batch(v2(real), v2(fake), eps, v2(comb), ppi, B)             # :31
            self._flush(disc)                                                     # previous iteration's D update lands here
            disc.set_high_tm(comb, B)
            disc.forward(B, training=True)                                        # :32-34
            dcomb = disc.backward(B, ones, need_wgrad=False)                      # :35
            ops.sumsq_batch_ch(v2(dcomb), ppi, T, B, gsq)                         # :36
            gnorm = torch.sqrt(gsq[:, :ch])
            gradient_reg = self.GAMMA * ((gnorm - 1.0) ** 2).mean()               # :37  (a constant w.r.t. D's weights)
            disc.params.zero_grad()
            if d_loss_fn is not None:
                loss_value, real_mean, fake_mean = self._critic_coupled(d_loss_fn, B, real, fake, noisy, sw_mean, low, sample_weight)
                disc_loss = loss_value + gradient_reg                             # :44-45 (regularization_losses)
                dscale = self._reduce_and_step(disc, d_opt)                       # :46-47
                continue
            noise.normal_into(v2(noisy[..., :ch]), self.noise_std, add=v2(real[..., :ch]))   # :40
            disc.set_high_tm(noisy, B)
            real_mean = disc.forward(B, training=True).mean()                     # :41
            disc.backward(B, self._const("real", B, -sw_mean / B), need_wgrad=True, need_input_grad=False)
            noise.normal_into(v2(noisy[..., :ch]), self.noise_std, add=v2(fake[..., :ch]))   # :42
            disc.set_high_tm(noisy, B)
            fake_mean = disc.forward(B, training=True).mean()                     # :43
            disc.backward(B, self._const("fake", B, sw_mean / B), need_wgrad=True, need_input_grad=False)
            disc_loss = (fake_mean - real_mean) * sw_mean + gradient_reg          # :44-45, train.py:11-12
            dscale = self._reduce_and_step(disc, d_opt)                           # :46-47
        if not pipelined:
            gen.params.zero_grad()                                                # generator step, :50-61
            self._gen_noise(low, B)
            fake = gen.forward(B, training=True, need_backward=True)              # overlaps the last D exchange
        self._flush(disc)
        d_gradient_param = self._grad_param_metric(disc, dscale)

        disc.set_high_tm(fake, B)
        gen_disc_loss = -disc.forward(B, training=True).mean()                    # :54
        reco_loss, dreco, n_rep = None, None, 1
        if reconstruction_loss is not None:                                       # :57-59
            # gen_loss = gen_disc_loss + reco_loss; a non-scalar reco_loss (e.g. metrics.wind_speed_weighted_rmse: one value per
            # sample) makes gen_loss non-scalar and tape.gradient (:60) differentiates the SUM of its elements: the adversarial
            # term then counts once per element
            reco_loss, dreco = self._reco_grad(reconstruction_loss, low, fake, B, T)
            n_rep = max(1, reco_loss.numel())
        dfake = disc.backward(B, self._const("gstep", B, -float(n_rep) / B), need_wgrad=False)
        if dreco is not None:
            ops.copy_channels(dreco, dfake[..., :ch], accumulate=True)
        if pipelined:
            # metrics recompute, :63-65: D(real) in inference mode reads neither the generator nor anything its backward
            # writes — on the second stream under the generator's backward pass (the discriminator's activations are free:
            # its last backward has produced dfake)
            gs, main = self._gen_stream, torch.cuda.current_stream(ops.device)
            gs.wait_stream(main)
            with torch.cuda.stream(gs):
                disc.set_high_tm(real, B)
                real_mean = disc.forward(B, training=False).mean()
                real_mean.record_stream(main)
        gen.backward(B, dfake)
        gscale = self._reduce_and_step(gen, g_opt)
        if pipelined:
            main.wait_stream(gs)
        else:
            disc.set_high_tm(real, B)                                             # metrics recompute, :63-68
            real_scores = disc.forward(B, training=False).clone()                 # overlaps the generator's exchange
            real_mean = real_scores.mean()
        self._flush(gen)
        g_gradient_param = self._grad_param_metric(gen, gscale)
        self._gen_noise(low, B)
        fake = gen.forward(B, training=False)
        disc.set_high_tm(fake, B)
        fake_scores = disc.forward(B, training=False)
        fake_mean = fake_scores.mean()
        self.last_fake_tm = fake
        if d_loss_fn is None:
            d_loss = (fake_mean - real_mean) * sw_mean                            # :67, train.py:11-12
        else:
            d_loss = self._compiled_loss(d_loss_fn(real_scores.view(B, 1), fake_scores.view(B, 1)), sample_weight, sw_mean)    # :67 compiled_loss = the custom callable
        return self._reduce_metrics({
            "g_loss": -fake_mean,
            "g_disc_loss": gen_disc_loss,
            "g_reco_loss": reco_loss,
            "d_loss": d_loss,
            "d_gradient_pen": gnorm.mean(),
            "g_gradient_param": g_gradient_param,
            "d_gradient_param": d_gradient_param,
            "_d_loss_train": disc_loss,
            "_d_real": real_mean,
            "_d_fake": fake_mean,
        })

    def _reco_grad(self, reconstruction_loss, low, fake_tm, B, T):
        """User-supplied reconstruction loss (train.py:19-26) evaluated with torch autograd on a leaf
        copy of the generator output; returns (loss, d loss / d fake as a time-major view)."""
        gen = self.gen
        ch = self.disc.ch
        fake_api = torch.empty(B, T, gen.S, gen.S, ch, dtype=fake_tm.dtype, device=fake_tm.device)
        gen.from_time_major(fake_tm, fake_api)
        leaf = fake_api.detach().requires_grad_(True)
        loss = reconstruction_loss(low[..., :2], leaf)
        (gapi,) = torch.autograd.grad(loss.sum(), leaf)
        gtm = self._buf("dreco", T * B, gen.S, gen.S, ch)
        gen.to_time_major(gapi, gtm)
        return loss.detach(), gtm

    def test_step(self, low, high, d_loss_fn=None):
        """GAN.test_step (ganbase.py:96-113): the discriminator's compiled loss (no sample weights, :103) on real vs
        generated, inference mode.  Draw order as the reference: the generator noise first (:99)."""
        gen, disc = self.gen, self.disc
        B, T = low.shape[0], low.shape[1]
        S, chp = gen.S, round4(disc.ch)
        gen.set_image(low)
        disc.set_low(low)
        real = self._buf("real", T * B, S, S, chp)
        gen.to_time_major(high, real)
        self._gen_noise(low, B)                                                   # :99
        disc.set_high_tm(real, B)
        real_scores = disc.forward(B, training=False).clone()                     # :100
        fake = gen.forward(B, training=False)                                     # :101
        disc.set_high_tm(fake, B)
        fake_scores = disc.forward(B, training=False)                             # :102
        self.last_fake_tm = fake
        if d_loss_fn is None:
            loss = fake_scores.mean() - real_scores.mean()                        # :103, train.py:11-12
        else:
            loss = d_loss_fn(real_scores.view(B, 1), fake_scores.view(B, 1))
        return self._reduce_metrics({"loss": loss})
