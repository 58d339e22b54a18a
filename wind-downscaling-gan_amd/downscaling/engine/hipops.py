"""HIP operator backend: thin typed wrappers over libwdgan.so for torch (ROCm) tensors.

PyTorch is used for device memory, streams and tensor views only; every arithmetic op below is a
hand-written gfx950 kernel reached through the C ABI (include/wdgan.h).  All activation tensors are
channels-last fp32 views with unit channel stride; 4-D views are (n_img, H, W, C), 2-D views (P, C).
"""
import ctypes as C
import os
import torch

from . import native
from .common import ConvGeom  # noqa: F401  (re-exported)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _v2(t):
    """(ptr, ld) of a 2-D [P, C] view with unit channel stride."""
    assert t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1), (t.shape, t.stride())
    ld = t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])
    return t.data_ptr(), ld


def _v4(t):
    """(ptr, ld, img_stride) of a 4-D [N, H, W, C] view: dense over (H, W) with pixel stride ld."""
    assert t.dim() == 4 and (t.shape[3] == 1 or t.stride(3) == 1), (t.shape, t.stride())
    n, h, w, c = t.shape
    ld = t.stride(2) if w > 1 else (t.stride(1) if h > 1 else max(c, t.stride(2)))
    if h > 1:
        assert t.stride(1) == w * ld, (t.shape, t.stride())
    isx = t.stride(0) if n > 1 else h * w * ld
    return t.data_ptr(), ld, isx


class PackedWeights:
    """Master HWIO weights (a view into the flat parameter buffer) + the two kernel layouts."""

    def __init__(self, ops, w):
        kh, kw, cin, cout = w.shape
        self.ops, self.w = ops, w
        self.taps, self.cin, self.cout = kh * kw, cin, cout
        cin_p, cout_p = (cin + 3) // 4 * 4, (cout + 3) // 4 * 4
        self.wF = torch.empty(cout * self.taps * cin_p, dtype=torch.float32, device=w.device)
        self.wD = w if cout % 4 == 0 else torch.empty(self.taps * cin * cout_p, dtype=torch.float32, device=w.device)
        self._half = {}
        self._hver = 0
        self._bf16_stale = True
        self._up4 = None
        self.refresh()

    def up4(self):
        """Composite 4x4 kernels of the fused bilinear-x2 + 5x5 transposed conv (csrc/upconv4.hip), rebuilt lazily
        after the master weights changed."""
        lib = self.ops.lib
        if self._up4 is None:
            self._up4 = torch.empty(int(lib.wdg_upconv4_weight_floats(self.cin, self.cout)), dtype=torch.float32,
                                    device=self.w.device)
            self._up4_stale = True
        if self._up4_stale:
            native.check(lib.wdg_upconv4_pack(self.w.data_ptr(), self.cin, self.cout, self._up4.data_ptr(),
                                              self.ops.stream), "upconv4_pack")
            self._up4_stale = False
        return self._up4

    def bf16(self):
        return self.half("bf16")

    def half(self, fmt="bf16"):
        """(wF16, wD16): 16-bit copies (fmt "bf16" or "fp16") of the packed layouts for the inference-precision
        kernels, converted lazily and again after the master weights changed."""
        cache = self._half.setdefault(fmt, {"wF": None, "wD": None, "ver": -1})
        tdt = torch.bfloat16 if fmt == "bf16" else torch.float16
        if cache["wF"] is None:
            cache["wF"] = torch.empty(self.wF.numel(), dtype=tdt, device=self.w.device)
            cache["wD"] = torch.empty(self.wD.numel(), dtype=tdt, device=self.w.device)
        if self._bf16_stale:            # the packed fp32 layouts changed since the last conversion of any format
            self._hver += 1
            self._bf16_stale = False
        if cache["ver"] != self._hver:
            lib, st = self.ops.lib, self.ops.stream
            conv = lib.wdg_convert_bf16 if fmt == "bf16" else lib.wdg_convert_f16
            native.check(conv(self.wF.data_ptr(), cache["wF"].data_ptr(), self.wF.numel(), st), "convert_16")
            native.check(conv(self.wD.data_ptr(), cache["wD"].data_ptr(), self.wD.numel(), st), "convert_16")
            cache["ver"] = self._hver
        return cache["wF"], cache["wD"]

    def interleaved(self, F):
        """wF with gate-interleaved rows for the ConvLSTM step in the implicit GEMM's epilogue (wdg_convlstm_step_gemm): row n =
        gate n & 3 of feature n >> 2.  A gathered copy, rebuilt lazily after the master weights changed."""
        cache = self.__dict__.setdefault("_il", {"buf": None, "ver": -1, "perm": None})
        if self._bf16_stale:            # (shares the staleness counter of the 16-bit copies)
            self._hver += 1
            self._bf16_stale = False
        if cache["ver"] != self._hver:
            L = self.wF.numel() // self.cout
            if cache["perm"] is None:
                n = torch.arange(4 * F, device=self.wF.device)
                cache["perm"] = (n & 3) * F + (n >> 2)
            src = self.wF.view(self.cout, L)
            if cache["buf"] is None:
                cache["buf"] = torch.empty_like(src)
            torch.index_select(src, 0, cache["perm"], out=cache["buf"])
            cache["ver"] = self._hver
        return cache["buf"]

    def lstm16(self):
        """(wl_fwd, wl_bwd): the recurrent kernel [3][3][16][64] in the LDS layouts of the 16-feature step kernels
        (wdg_convlstm16_pack), rebuilt lazily after the master weights changed."""
        cache = self.__dict__.setdefault("_l16", {"f": None, "b": None, "ver": -1})
        if self._bf16_stale:            # (shares the staleness counter of the 16-bit copies)
            self._hver += 1
            self._bf16_stale = False
        if cache["ver"] != self._hver:
            if cache["f"] is None:
                cache["f"], cache["b"] = torch.empty(9216, device=self.w.device), torch.empty(9216, device=self.w.device)
            native.check(self.ops.lib.wdg_convlstm16_pack(self.w.data_ptr(), cache["f"].data_ptr(), cache["b"].data_ptr(),
                                                          self.ops.stream), "convlstm16_pack")
            cache["ver"] = self._hver
        return cache["f"], cache["b"]

    def as_1x1(self):
        """The same weights as a 1x1 convolution with taps*cin input channels: w[t][i][o] viewed as [t*cin + i][o].
        Both kernel layouts coincide with this object's (wF is [cout][taps][cin], wD is w itself), so the view shares
        them and needs no packing of its own (cin % 4 == 0 and cout % 4 == 0 only)."""
        assert self.cin % 4 == 0 and self.cout % 4 == 0 and self.wD is self.w
        sub = object.__new__(PackedWeights)
        sub.ops, sub.taps, sub.cin, sub.cout = self.ops, 1, self.taps * self.cin, self.cout
        sub.w = self.w.view(1, 1, self.taps * self.cin, self.cout)
        sub.wF, sub.wD = self.wF, sub.w
        return sub

    def column_slice(self, n0, n1):
        """Forward-only view of output channels [n0, n1): the rows of wF are contiguous per output channel."""
        sub = object.__new__(PackedWeights)
        sub.ops, sub.w, sub.taps, sub.cin, sub.cout = self.ops, None, self.taps, self.cin, n1 - n0
        ld = self.taps * ((self.cin + 3) // 4 * 4)
        sub.wF = self.wF[n0 * ld:n1 * ld]
        sub.wD = None
        sub._parent, sub._range = self, (n0 * ld, n1 * ld)
        return sub

    def mark_stale(self):
        self._bf16_stale = True

    def refresh(self):
        self.mark_stale()
        self._up4_stale = True
        lib = self.ops.lib
        native.check(lib.wdg_weight_pack(self.w.data_ptr(), self.wF.data_ptr(),
                                         0 if self.wD is self.w else self.wD.data_ptr(),
                                         self.taps, self.cin, self.cout, self.ops.stream), "weight_pack")


class _PrepBatch:
    def __init__(self, ops, entries):
        self.ops, self.entries = ops, entries
        arr = (native.PrepLayer * len(entries))()
        for i, (pk, u) in enumerate(entries):
            kh_kw_cin, cout = pk.w.numel() // pk.cout, pk.cout
            arr[i] = native.PrepLayer(pk.w.data_ptr(), _ptr(u), pk.wF.data_ptr(), 0 if pk.wD is pk.w else pk.wD.data_ptr(),
                                      kh_kw_cin, cout, pk.taps, pk.cin, pk.cout, int(u is not None))
        self.handle = C.c_void_p()
        native.check(ops.lib.wdg_prep_batch_create(C.byref(self.handle), arr, len(entries)), "prep_batch_create")
        self.scratch = ops.empty(max(int(ops.lib.wdg_prep_batch_scratch_floats(self.handle)), 4))
        self.any_sn = any(u is not None for _, u in entries)
        self.epoch = 0      # bumped whenever run() rewrote weights / packed layouts (key of captured inference graphs)

    def run(self, sn, pack_all):
        flags = (native.PREP_SN if sn and self.any_sn else 0) | (native.PREP_PACK_ALL if pack_all else 0)
        if not flags:
            return
        native.check(self.ops.lib.wdg_prep_batch_run(self.handle, self.scratch.data_ptr(), flags, self.ops.stream),
                     "prep_batch_run")
        self.epoch += 1
        for pk, u in self.entries:
            if pack_all or (sn and u is not None):
                pk.mark_stale()
                pk._up4_stale = True

    def __del__(self):
        try:
            self.ops.lib.wdg_prep_batch_destroy(self.handle)
        except Exception:
            pass


class _Fork:
    def __init__(self, ops, name="side", stream=None):
        self.ops = ops
        if stream is not None:
            self.side = stream          # (a stream the caller placed: HipOps.concurrent_streams)
        else:
            streams = ops.__dict__.setdefault("_side_streams", {})
            if name not in streams:
                streams[name] = torch.cuda.Stream(device=ops.device)
            self.side = streams[name]
        self.ctx = None

    def __enter__(self):
        self.side.wait_stream(torch.cuda.current_stream(self.ops.device))
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self.ctx.__exit__(*exc)
        return False

    def join(self):
        torch.cuda.current_stream(self.ops.device).wait_stream(self.side)


class HipOps:
    name = "hip"
    dtype = torch.float32
    supports_graphs = True     # launches go to torch's current stream, so torch.cuda.graph captures them
    supports_streams = True    # ... and work can be forked onto side streams (fork(), the trainer's multi-stream schedule)

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise native.NativeError("HipOps needs a ROCm GPU (torch.cuda.is_available() is False); "
                                     "there is no CPU execution path in this package")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        torch.cuda.set_device(self.device)
        self.lib = native.load()
        self._plans = {}
        self._ws = None
        self._sn_scratch = None
        self.upconv4 = True   # fused upsample + 5x5 transposed conv through the composite-kernel path
        self.upconv_col = True   # its backward in column form on the low-res grid
        self.z16 = os.environ.get("WDG_Z16", "0") == "1"   # 16-bit inference: the column GEMM's result z in the operand format
        self.upconv_colfwd = os.environ.get("WDG_UPCONV_COLFWD", "1") != "0"   # forward in column form (1x1 GEMM + bilinear gather): 1.22 vs 1.6 ms
        self.upconv_fused16 = os.environ.get("WDG_UPCONV_FUSED16", "1") != "0"   # 16-bit inference: column GEMM + gather in one launch
        # the 16-channel activation between the generator's last two layers in the 16-bit operand format (inference precision)
        # WDG_CHAIN_TIMING=1: [count, seconds] of the host inside HipOps.chain's graph replays (tools: who keeps the streams waiting)
        self.chain_timing = [0, 0.0] if os.environ.get("WDG_CHAIN_TIMING", "0") != "0" else None
        self.act16 = os.environ.get("WDG_ACT16", "1") != "0"
        # generator input ([image | noise]) assembled by one kernel when the noise is drawn on the device (LazyNoise)
        self.input_fused = os.environ.get("WDG_INPUT_FUSED", "1") != "0"
        self._scratch_bufs = {}
        # per-timestep launch chains (the ConvLSTM time loops at n_timesteps > 1) replayed from captured HIP graphs: see chain()
        self.gates_x = os.environ.get("WDG_GATES_X", "1") != "0"      # T > 1: the 5 -> 16 ConvLSTM's input convolution in its own kernel
        self.chain_graphs = os.environ.get("WDG_CHAIN_GRAPHS", "1") != "0"
        self._chains = {}
        self._capture_ws = None      # chain(): scratch dict of the graph being captured

    def chain(self, key, fn, graphs=None):
        """Runs fn() — a chain of dependent per-timestep launches on FIXED buffers with no host-side effects (a ConvLSTM's time
        loop) — from a HIP graph.  At the shipped sequence length a train step issues ~1,650 such launches of 8-45 us each;
        through Python / ctypes the host needs ~30 us per launch, so it cannot run ahead of the device and the streams that
        should overlap (generator / discriminator / twin discriminator) are fed one after the other.  The first call with a key
        runs eagerly (plans and scratch come into being), the second is captured, later ones are one graph launch.
        `key` must name every buffer address and shape the chain touches — activations AND the lazily rebuilt weight layouts
        the step kernels read (convlstm_step_prepare returns those); the library's tuning epoch is added here.
        `graphs`: the dict that owns the captured graphs — the caller's (a layer's), so the graphs die with the buffers whose
        addresses they hold; without one, a process-wide dict.  Split-K scratch requested while capturing is allocated for
        that graph alone (see _workspace): graphs replayed at the same time on different streams never share scratch."""
        if not self.chain_graphs or torch.cuda.is_current_stream_capturing():
            return fn()
        if graphs is None:
            graphs = self._chains
        key = key + (int(self.lib.wdg_tuning_epoch()),)
        entry = graphs.get(key)
        if entry is None:
            seen = graphs.setdefault("seen", {})
            seen[key] = seen.get(key, 0) + 1
            if seen[key] < 2:
                return fn()
            own_ws = {}
            self._capture_ws = own_ws
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    fn()
            except Exception as exc:              # capture not possible here: stay eager from now on, and say so
                torch.cuda.synchronize()
                self.chain_graphs = False
                import warnings
                warnings.warn(f"HipOps.chain: HIP graph capture failed ({type(exc).__name__}: {exc}); "
                              "time loops run as eager launches from now on", RuntimeWarning)
                return fn()
            finally:
                self._capture_ws = None
            if len(graphs) > 64:                  # (keys of replaced buffers / older tuning epochs)
                for k in [k for k in graphs if k != "seen"]:
                    del graphs[k]
            seen.pop(key, None)
            entry = graphs[key] = (graph, own_ws)  # (the graph's private scratch lives exactly as long as the graph)
        if self.chain_timing is None:
            entry[0].replay()
            return
        import time
        t0 = time.perf_counter()
        entry[0].replay()
        self.chain_timing[0] += 1
        self.chain_timing[1] += time.perf_counter() - t0

    # ---- plumbing ---------------------------------------------------------------------------
    @property
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def from_host(self, arr):
        return torch.as_tensor(arr, dtype=torch.float32).to(self.device)

    def pack_weights(self, w):
        return PackedWeights(self, w)

    def _workspace(self, nbytes):
        """Split-K / split-pixel scratch of the CURRENT stream (work forked onto a side stream gets its own)."""
        if self._capture_ws is not None:
            # inside chain()'s capture: the captured kernels bake this address in, and the graph may later be replayed on any
            # stream beside other graphs — the scratch belongs to this graph alone
            ws = self._capture_ws.get("ws")
            if ws is None or ws.numel() < nbytes:
                assert ws is None or nbytes == 0, "chain capture: scratch grew between launches of one chain"
                ws = self._capture_ws["ws"] = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.device)
            return ws
        if self._ws is None:
            self._ws = {}
        key = self.stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = self._ws[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=self.device)
        return ws

    def concurrent_streams(self, n):
        """`n` HIP streams that PROVABLY run beside torch's current stream and beside each other — chosen by measurement.

        Why not just torch.cuda.Stream(): HIP multiplexes its streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default
        4) in creation order, torch hands out streams of a 32-entry pool round-robin to every caller, and two streams that
        land on one hardware queue run one after the other.  Whether the trainer's generator / twin-discriminator streams
        overlap with the main stream therefore depended on who had taken pool entries before — e.g. creating an RCCL process
        group shifted the assignment and cost the data-parallel step its overlap (70.9 vs 67.9 ms at the headline shape with
        the group merely initialised, profiles/r04c_dp.txt; raising the queue count instead made cross-queue waits expensive:
        profiles/r04d_queues.txt, r04e_queues.txt).  The probe: a single-workgroup spin kernel on every stream of the
        candidate set at once takes one spin time when they sit on different hardware queues and one per stream when they
        share one.  Falls back to plain pool streams when no concurrent set exists (one hardware queue)."""
        cached = self.__dict__.setdefault("_cstreams", [])
        if len(cached) >= n:
            return cached[:n]
        main = torch.cuda.current_stream(self.device)
        cycles = 400_000

        def spin_ms(streams):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(self.device)
            e0.record(main)
            for st in streams:                # every stream starts behind e0 ... (all waits BEFORE any spin is enqueued)
                if st is not main:
                    st.wait_stream(main)
            for st in streams:                # ... and spins once
                with torch.cuda.stream(st):
                    torch.cuda._sleep(cycles)
            for st in streams:
                if st is not main:
                    main.wait_stream(st)
            e1.record(main)
            torch.cuda.synchronize(self.device)
            return e0.elapsed_time(e1)

        spin_ms([main])
        t1 = min(spin_ms([main]) for _ in range(3))
        chosen = list(cached)
        tried = 0
        while len(chosen) < n and tried < 24:
            cand = torch.cuda.Stream(device=self.device)
            tried += 1
            group = [main] + chosen + [cand]
            t = min(spin_ms(group) for _ in range(2))
            if t < 1.5 * t1:                  # all of them side by side (a shared queue costs at least 2 x t1)
                chosen.append(cand)
        found = len(chosen)
        while len(chosen) < n:                # no concurrent set (e.g. one hardware queue): still correct, just serial
            chosen.append(torch.cuda.Stream(device=self.device))
        self._cstreams = chosen
        self._cstreams_probe = {"spin_ms": t1, "candidates_tried": tried, "concurrent_found": found}
        return chosen[:n]

    def fork(self, name="side", stream=None):
        """Context manager: run the enclosed launches on the side stream `name` that starts after everything enqueued so far
        on the current stream; `join()` on the returned object makes the current stream wait for them.  Used for chains of
        small launches (the per-timestep recurrent kernels at T > 1) and for work off the critical path of a pass (the
        weight gradients of a backward pass: name "wgrad").  stream: run on THIS stream instead of the named pool stream."""
        return _Fork(self, name, stream)

    def _plan(self, x, y, cin, cout, g: ConvGeom, w_ld=0):
        px, ldx, isx = _v4(x)
        py, ldy, isy = _v4(y)
        n, H, W, _ = x.shape
        _, Ho, Wo, _ = y.shape
        return self._plan_dims(n, H, W, cin, ldx, isx, Ho, Wo, cout, ldy, isy, g, w_ld)

    supports_weight_slices = True

    def weight_slices_ok(self, x, dy_slice, pk, g):
        """Can conv_dgrad_slice / conv_wgrad_slice run this layer on a channel range of width dy_slice.shape[-1]?
        wdg_conv_plan_create_sliced accepts only geometries whose three directions run on the implicit-GEMM kernels (it
        returns WDG_ERR_ARG where the halo / thin kernels would be picked: stride 1, 3x3, few channels, large maps); the
        caller then keeps the full-width conv_wgrad / conv_dgrad."""
        try:
            self._plan(x, dy_slice, pk.cin, dy_slice.shape[-1], g, w_ld=pk.cout)
            return True
        except native.NativeError:
            return False

    def conv_dgrad_slice(self, dy, pk, n0, n1, dx, g, accumulate=False):
        """dx (+)= conv_transpose(dy, W[..., n0:n1]) for a channel RANGE of the layer's output: dy is the [.., n0:n1] view of the
        output-gradient tensor, pk the pack of the FULL layer (wdg_conv_plan_create_sliced).  A ConvLSTM2D at n_timesteps = 1 has
        a dead forget gate (c_0 = 0): its quarter of the reduction is skipped this way."""
        assert pk.cout % 4 == 0 and n0 % 4 == 0 and n1 % 4 == 0 and dy.shape[-1] == n1 - n0
        plan, wsb, _ = self._plan(dx, dy, pk.cin, n1 - n0, g, w_ld=pk.cout)
        ws = self._workspace(wsb)
        native.check(self.lib.wdg_conv_dgrad(plan, dy.data_ptr(), pk.wD.data_ptr() + 4 * n0, None, dx.data_ptr(), 0, 0.2,
                                             int(accumulate), ws.data_ptr(), ws.numel(), self.stream), "conv_dgrad(slice)")

    def conv_wgrad_slice(self, x, dy, pk, n0, n1, dw, g, accumulate=True):
        """dw[..., n0:n1] (+)= x (*) dy for a channel range (see conv_dgrad_slice); dw is the FULL [kh,kw,Cin,Cout] gradient."""
        assert dw.is_contiguous() and dw.shape[-1] == pk.cout and pk.cout % 4 == 0 and n0 % 4 == 0 and dy.shape[-1] == n1 - n0
        plan, wsb, _ = self._plan(x, dy, pk.cin, n1 - n0, g, w_ld=pk.cout)
        ws = self._workspace(wsb)
        native.check(self.lib.wdg_conv_wgrad(plan, x.data_ptr(), dy.data_ptr(), dw.data_ptr() + 4 * n0, int(accumulate),
                                             ws.data_ptr(), ws.numel(), self.stream), "conv_wgrad(slice)")

    def _plan_dims(self, n, H, W, cin, ldx, isx, Ho, Wo, cout, ldy, isy, g, w_ld=0):
        key = (n, H, W, cin, ldx, isx, Ho, Wo, cout, ldy, isy, g.kh, g.kw, g.stride, g.pad) + ((w_ld,) if w_ld else ())
        plan = self._plans.get(key)
        if plan is None:
            geom = native.ConvGeom(n, H, W, cin, ldx, isx, Ho, Wo, cout, ldy, isy, g.kh, g.kw, g.stride, g.pad, g.pad)
            handle = C.c_void_p()
            if w_ld:
                native.check(self.lib.wdg_conv_plan_create_sliced(C.byref(handle), C.byref(geom), int(w_ld)), f"plan_create_sliced{key}")
            else:
                native.check(self.lib.wdg_conv_plan_create(C.byref(handle), C.byref(geom)), f"plan_create{key}")
            info = (C.c_int32 * 8)()
            native.check(self.lib.wdg_conv_plan_info(handle, info), "plan_info")
            plan = (handle, int(self.lib.wdg_conv_ws_bytes(handle)), tuple(info))
            self._plans[key] = plan
        return plan

    @staticmethod
    def _label(bm, bn):
        return "wdg_conv_halo_kernel<%d>" % (bn // 16) if bm == 0 else "wdg_igemm_kernel<%d,%d>" % (bm, bn)

    def conv_kernel_label(self, which, x, y, pk, g, cout=None, w_ld=0):
        """Name of the kernel template a conv call launches (profiling labels; which: fwd|dgrad|wgrad).  cout / w_ld: the
        channel-range calls (conv_dgrad_slice / conv_wgrad_slice)."""
        info = self._plan(x, y, pk.cin, pk.cout if cout is None else cout, g, w_ld=w_ld)[2]
        if which == "fwd":
            return self._label(info[0], info[1])
        if which == "dgrad":
            return self._label(info[3], info[4])
        if info[6] < 0:
            return "wdg_wgrad_thin_kernel"
        return "wdg_wgrad_halo_kernel" if info[6] == 0 else "wdg_wgrad_kernel<%d>" % info[6]

    # ---- convolution family -----------------------------------------------------------------
    def conv_fwd(self, x, pk, bias, y, g, act=False, accumulate=False, slope=0.2, bn_stats=None, bn_affine=None):
        """y = act(conv(x, W) + bias) (+ y);  x:(N,H,W,>=Cin) y:(N,Ho,Wo,>=Cout).
        bn_stats [R, 2*Cout] fp64 (zeroed by the caller): the launch also accumulates the per-channel sum / sum of squares
        of y there (the first pass of a training-mode BatchNormalization on y); bn_affine [2*Cout]: y is additionally
        scaled / shifted per channel (inference-mode BatchNormalization) — see wdg_conv_fwd_bn."""
        plan, wsb, _ = self._plan(x, y, pk.cin, pk.cout, g)
        ws = self._workspace(wsb)
        if bn_stats is not None or bn_affine is not None:
            assert not accumulate and (bn_stats is None or (bn_stats.dtype == torch.float64 and bn_stats.shape[1] == 2 * ((pk.cout + 3) // 4 * 4)))
            native.check(self.lib.wdg_conv_fwd_bn(plan, x.data_ptr(), pk.wF.data_ptr(), _ptr(bias), y.data_ptr(), int(act), slope,
                                                  _ptr(bn_stats), bn_stats.shape[0] if bn_stats is not None else 0,
                                                  _ptr(bn_affine), ws.data_ptr(), ws.numel(), self.stream), "conv_fwd_bn")
            return
        native.check(self.lib.wdg_conv_fwd(plan, x.data_ptr(), pk.wF.data_ptr(), _ptr(bias), y.data_ptr(),
                                           int(act), slope, int(accumulate), ws.data_ptr(), ws.numel(),
                                           self.stream), "conv_fwd")

    def conv_fwd_ln(self, x, pk, bias, y, z, g, gamma, beta, eps, mean_rstd, act=True, slope=0.2):
        """y = act(conv(x, W) + bias); z = LayerNorm(y) over the channels; mean_rstd [P, 2] — wdg_conv_fwd_ln (the norm runs in
        the epilogue that owns complete rows, or as the standalone pass behind the conv)."""
        plan, wsb, _ = self._plan(x, y, pk.cin, pk.cout, g)
        ws = self._workspace(wsb)
        assert z.shape[:3] == y.shape[:3] and z.shape[3] == pk.cout
        _, ldz, isz = _v4(z)            # z may live in a wider buffer (a channel slice of a concatenation)
        native.check(self.lib.wdg_conv_fwd_ln_strided(plan, x.data_ptr(), pk.wF.data_ptr(), _ptr(bias), y.data_ptr(), z.data_ptr(),
                                                      ldz, isz, gamma.data_ptr(), beta.data_ptr(), eps, _ptr(mean_rstd), int(act),
                                                      slope, ws.data_ptr(), ws.numel(), self.stream), "conv_fwd_ln")

    def conv_dgrad(self, dy, pk, dx, g, bias=None, act=False, accumulate=False, slope=0.2, bn_stats=None, bn_affine=None):
        """dx = act(conv_transpose(dy, W) + bias) (+ dx);  the geometry is that of the forward conv.  bn_stats / bn_affine
        as in conv_fwd, over the Cin channels this launch writes."""
        plan, wsb, _ = self._plan(dx, dy, pk.cin, pk.cout, g)
        ws = self._workspace(wsb)
        if bn_stats is not None or bn_affine is not None:
            assert not accumulate and (bn_stats is None or (bn_stats.dtype == torch.float64 and bn_stats.shape[1] == 2 * ((pk.cin + 3) // 4 * 4)))
            native.check(self.lib.wdg_conv_dgrad_bn(plan, dy.data_ptr(), pk.wD.data_ptr(), _ptr(bias), dx.data_ptr(), int(act), slope,
                                                    _ptr(bn_stats), bn_stats.shape[0] if bn_stats is not None else 0,
                                                    _ptr(bn_affine), ws.data_ptr(), ws.numel(), self.stream), "conv_dgrad_bn")
            return
        native.check(self.lib.wdg_conv_dgrad(plan, dy.data_ptr(), pk.wD.data_ptr(), _ptr(bias), dx.data_ptr(),
                                             int(act), slope, int(accumulate), ws.data_ptr(), ws.numel(),
                                             self.stream), "conv_dgrad")

    def conv_dgrad_lnbwd(self, dy, pk, dx, g, y, mean_rstd, gamma, c0, C, act_slope, dgamma, dbeta, dbias, par_ws=None):
        """dx = conv_transpose(dy, W), then the LayerNorm + LeakyReLU backward of the layer that PRODUCED channels [c0, c0 + C) of
        this convolution's input, applied to those channels of dx in place (dx[..., c0:c0+C] becomes the gradient w.r.t. that
        producer's pre-activation) — wdg_conv_dgrad_lnbwd: in the data gradient's epilogue where a tile owns complete pixels
        (the standalone wdg_ln_bwd pass over dz disappears), conv_dgrad + ln_bwd otherwise.  y [N,H,W,C]: the norm's input;
        par_ws: zeroed scratch from lnbwd_scratch(C), needed with parameter gradients (one per concurrent caller)."""
        plan, wsb, _ = self._plan(dx, dy, pk.cin, pk.cout, g)
        ws = self._workspace(wsb)
        py, ldy, isy = _v4(y)
        native.check(self.lib.wdg_conv_dgrad_lnbwd(plan, dy.data_ptr(), pk.wD.data_ptr(), dx.data_ptr(), py, ldy, isy, mean_rstd.data_ptr(),
                                                   gamma.data_ptr(), int(c0), int(C), float(act_slope), _ptr(dgamma), _ptr(dbeta), _ptr(dbias),
                                                   _ptr(par_ws), ws.data_ptr(), ws.numel(), self.stream), "conv_dgrad_lnbwd")

    def conv_dgrad_lnbwd_route(self, dy, pk, dx, g, y, c0, C, with_param_grads=True):
        """Which kernels a conv_dgrad_lnbwd call launches (profiling labels): 0 data gradient + ln_bwd, 1 the implicit-GEMM
        epilogue, 2 the 7x7 stride-3 patch kernel (csrc/dgrad_patch_s3.hip)."""
        plan, wsb, _ = self._plan(dx, dy, pk.cin, pk.cout, g)
        _, ldy, _ = _v4(y)
        return int(self.lib.wdg_conv_dgrad_lnbwd_route(plan, int(c0), int(C), ldy, int(bool(with_param_grads)), self._workspace(wsb).numel()))

    def lnbwd_scratch(self, C):
        return self.zeros(int(self.lib.wdg_conv_dgrad_lnbwd_par_floats(int(C))))

    def _is16(self, t, fmt):
        if t.dtype in (torch.bfloat16, torch.float16):
            if t.dtype != self.H16_DTYPES[fmt]:
                raise ValueError(f"tensor is {t.dtype}, the operand format is {fmt}")
            return 1
        return 0

    def conv_fwd_bf16(self, x, pk, bias, y, g, act=False, affine=None, accumulate=False, slope=0.2, fmt="bf16"):
        """Inference precision: y = affine(act(conv(r16(x), r16(W)) + bias)), fp32 accumulation; fmt "bf16" | "fp16".
        x and / or y may be tensors in the 16-bit operand format (wdg_conv_fwd_h16_act16: the patch kernel only)."""
        plan, _, _ = self._plan(x, y, pk.cin, pk.cout, g)
        in16, out16 = self._is16(x, fmt), self._is16(y, fmt)
        if in16 or out16:
            assert not accumulate
            native.check(self.lib.wdg_conv_fwd_h16_act16(plan, x.data_ptr(), in16, pk.half(fmt)[0].data_ptr(), 0 if fmt == "bf16" else 1,
                                                         _ptr(bias), _ptr(affine), y.data_ptr(), out16, int(act), slope, self.stream),
                         "conv_fwd_h16_act16")
            return
        fn = self.lib.wdg_conv_fwd_bf16 if fmt == "bf16" else self.lib.wdg_conv_fwd_f16
        native.check(fn(plan, x.data_ptr(), pk.half(fmt)[0].data_ptr(), _ptr(bias), _ptr(affine),
                        y.data_ptr(), int(act), slope, int(accumulate), self.stream), "conv_fwd_16")

    # ---- tile bookkeeping of the inference driver (csrc/tiling.hip) ----------------------------------------------
    def tiles_gather_normalise(self, field, keys4, T, S):
        """field [Ttot, LAT, LON, C] fp32, keys4 [N, 4] int32 {sx, row0, k, 0} -> normalised tiles [N, T, S, S, C] (api.py:117-129)."""
        assert field.is_contiguous() and field.dtype == torch.float32 and keys4.dtype == torch.int32 and keys4.is_contiguous()
        _, LAT, LON, C = field.shape
        N = keys4.shape[0]
        tiles = torch.empty(N, T, S, S, C, dtype=torch.float32, device=field.device)
        R = 64
        scratch = torch.empty(R * S * C * 3, dtype=torch.float64, device=field.device)
        mean_std = torch.empty(S * C, 2, dtype=torch.float32, device=field.device)
        native.check(self.lib.wdg_tiles_gather_normalise(field.data_ptr(), LAT, LON, C, keys4.data_ptr(), N, T, S, tiles.data_ptr(),
                                                         scratch.data_ptr(), R, mean_std.data_ptr(), self.stream), "tiles_gather_normalise")
        return tiles

    def tiles_blend(self, pred, keys4, n_real, acc, cnt, crop):
        """acc[t, lat, lon, :] += pred tile values, cnt += 1 for the first n_real tiles of the group (api.py:139-150)."""
        B, T, S, _, ldp = pred.shape
        assert pred.is_contiguous() and acc.is_contiguous() and cnt.is_contiguous() and acc.dtype == torch.float64 and cnt.dtype == torch.int32
        _, LAT, LON, _ = acc.shape
        native.check(self.lib.wdg_tiles_blend(pred.data_ptr(), ldp, keys4.data_ptr(), n_real, T, S, crop, LAT, LON, acc.data_ptr(),
                                              cnt.data_ptr(), self.stream), "tiles_blend")

    lstm_step_gemm = os.environ.get("WDG_LSTM_STEP_GEMM", "1") != "0"   # the implicit-GEMM form of the fused step (A/B switch)

    def convlstm_step_supported(self, h_prev, gates_t, pk, g, F):
        """fp32 ConvLSTM recurrent step in one launch?  The halo-tile kernel with the cell update in its epilogue
        (wdg_convlstm_step: the discriminator's thin layers) or the implicit GEMM with gate-interleaved weight rows
        (wdg_convlstm_step_gemm: the generator's 128-feature layer)."""
        plan, _, _ = self._plan(h_prev, gates_t, pk.cin, pk.cout, g)
        if self.lib.wdg_convlstm_step_supported(plan, F):
            return True
        return bool(self.lstm_step_gemm and pk.cout == 4 * F and
                    self.lib.wdg_convlstm_step_gemm_supported(plan, F))

    def convlstm_step_prepare(self, h_prev, pk, gates_t, g, F):
        """Host-side, once before a time loop of convlstm_step calls: the lazily rebuilt weight layout of the GEMM form must be
        current BEFORE the loop — the loop itself may be replayed from a captured graph (chain).  Returns the device addresses of
        those layouts: they are part of the identity of a captured loop (chain's key)."""
        plan, _, _ = self._plan(h_prev, gates_t, pk.cin, pk.cout, g)
        if self._lstm16(plan, pk, F):
            f, b = pk.lstm16()
            return (f.data_ptr(), b.data_ptr())
        if not self.lib.wdg_convlstm_step_supported(plan, F):
            return (pk.interleaved(F).data_ptr(),)
        return ()

    def _lstm16(self, plan, pk, F):
        """The 16-feature recurrent steps in their own kernels (csrc/convlstm16.hip)?"""
        return F == 16 and pk.w is not None and tuple(pk.w.shape) == (3, 3, 16, 64) and pk.w.is_contiguous() and \
            bool(self.lib.wdg_convlstm16_supported(plan))

    def convlstm_step(self, h_prev, pk, gates_t, c_prev, c_out, h_out, g, F):
        """gates_t += conv(h_prev); c_out, h_out from the cell update (gates_t keeps the pre-activations for the backward)."""
        plan, _, _ = self._plan(h_prev, gates_t, pk.cin, pk.cout, g)
        _, ldc, _ = _v4(c_out)
        _, ldh, _ = _v4(h_out)
        assert _v4(c_prev)[1] == ldc
        if self._lstm16(plan, pk, F):
            native.check(self.lib.wdg_convlstm16_step(plan, h_prev.data_ptr(), pk.lstm16()[0].data_ptr(), gates_t.data_ptr(),
                                                      c_prev.data_ptr(), c_out.data_ptr(), ldc, h_out.data_ptr(), ldh, self.stream),
                         "convlstm16_step")
            return
        if not self.lib.wdg_convlstm_step_supported(plan, F):
            native.check(self.lib.wdg_convlstm_step_gemm(plan, h_prev.data_ptr(), pk.interleaved(F).data_ptr(), gates_t.data_ptr(),
                                                         c_prev.data_ptr(), c_out.data_ptr(), ldc, h_out.data_ptr(), ldh, F,
                                                         self.stream), "convlstm_step_gemm")
            return
        native.check(self.lib.wdg_convlstm_step(plan, h_prev.data_ptr(), pk.wF.data_ptr(), gates_t.data_ptr(), c_prev.data_ptr(),
                                                c_out.data_ptr(), ldc, h_out.data_ptr(), ldh, F, self.stream), "convlstm_step")

    def convlstm_bwd_step_supported(self, dh_prev, dgates_t, pk, g, F):
        plan, _, _ = self._plan(dh_prev, dgates_t, pk.cin, pk.cout, g)
        return pk.wD is not None and bool(self.lib.wdg_convlstm_bwd_step_supported(plan, F))

    def convlstm_bwd_step(self, dgates_next, pk, dh_prev, gates_t, c_prev, c_cur, dc_in, dgates_out, dc_out, g, F):
        """dh_prev += conv_transpose(dgates_next); then the cell backward of that timestep -> dgates_out, dc_out (may be None)."""
        plan, _, _ = self._plan(dh_prev, dgates_next, pk.cin, pk.cout, g)
        _, ldc, _ = _v4(c_cur)
        if self._lstm16(plan, pk, F):
            native.check(self.lib.wdg_convlstm16_bwd_step(plan, dgates_next.data_ptr(), pk.lstm16()[1].data_ptr(), dh_prev.data_ptr(),
                                                          gates_t.data_ptr(), _ptr(c_prev), c_cur.data_ptr(), dc_in.data_ptr(),
                                                          dgates_out.data_ptr(), _ptr(dc_out), ldc, self.stream), "convlstm16_bwd_step")
            return
        native.check(self.lib.wdg_convlstm_bwd_step(plan, dgates_next.data_ptr(), pk.wD.data_ptr(), dh_prev.data_ptr(), gates_t.data_ptr(),
                                                    _ptr(c_prev), c_cur.data_ptr(), dc_in.data_ptr(), dgates_out.data_ptr(), _ptr(dc_out),
                                                    ldc, F, self.stream), "convlstm_bwd_step")

    # ---- both recurrent layers of the discriminator in one launch per timestep (csrc/convlstm16.hip: wdg_convlstm16_pair_*) ----
    def convlstm_pair_supported(self, h16, gates16, pk16, h2, gates2, pk2, g):
        """h16 / gates16 / pk16: one timestep's views and the recurrent pack of the 16-feature layer; h2 / gates2 / pk2: the
        two-feature layer's."""
        if os.environ.get("WDG_LSTM_PAIR", "1") == "0":
            return False
        p16, _, _ = self._plan(h16, gates16, pk16.cin, pk16.cout, g)
        p2, _, _ = self._plan(h2, gates2, pk2.cin, pk2.cout, g)
        return self._lstm16(p16, pk16, 16) and pk2.wD is not None and bool(self.lib.wdg_convlstm16_pair_supported(p16, p2))

    def convlstm_pair_step(self, a, b, g):
        """a = (h_prev, pk, gates_t, c_prev, c_out, h_out) of the 16-feature layer, b the same of the two-feature layer: convlstm_step
        of both in one launch."""
        h1, pk1, g1, cp1, co1, ho1 = a
        h2, pk2, g2, cp2, co2, ho2 = b
        p16, _, _ = self._plan(h1, g1, pk1.cin, pk1.cout, g)
        p2, _, _ = self._plan(h2, g2, pk2.cin, pk2.cout, g)
        native.check(self.lib.wdg_convlstm16_pair_step(p16, h1.data_ptr(), pk1.lstm16()[0].data_ptr(), g1.data_ptr(), cp1.data_ptr(),
                                                       co1.data_ptr(), _v4(co1)[1], ho1.data_ptr(), _v4(ho1)[1],
                                                       p2, h2.data_ptr(), pk2.wF.data_ptr(), g2.data_ptr(), cp2.data_ptr(), co2.data_ptr(),
                                                       _v4(co2)[1], ho2.data_ptr(), _v4(ho2)[1], self.stream), "convlstm16_pair_step")

    def convlstm_pair_bwd_step(self, a, b, g):
        """a = (dgates_next, pk, dh_prev, gates_t, c_prev | None, c_cur, dc_in, dgates_out, dc_out | None) of the 16-feature layer, b of
        the two-feature layer: convlstm_bwd_step of both in one launch."""
        d1, pk1, dh1, g1, cp1, cc1, dci1, do1, dco1 = a
        d2, pk2, dh2, g2, cp2, cc2, dci2, do2, dco2 = b
        p16, _, _ = self._plan(dh1, d1, pk1.cin, pk1.cout, g)
        p2, _, _ = self._plan(dh2, d2, pk2.cin, pk2.cout, g)
        native.check(self.lib.wdg_convlstm16_pair_bwd_step(p16, d1.data_ptr(), pk1.lstm16()[1].data_ptr(), dh1.data_ptr(), g1.data_ptr(),
                                                           _ptr(cp1), cc1.data_ptr(), dci1.data_ptr(), do1.data_ptr(), _ptr(dco1), _v4(cc1)[1],
                                                           p2, d2.data_ptr(), pk2.wD.data_ptr(), dh2.data_ptr(), g2.data_ptr(), _ptr(cp2),
                                                           cc2.data_ptr(), dci2.data_ptr(), do2.data_ptr(), _ptr(dco2), _v4(cc2)[1],
                                                           self.stream), "convlstm16_pair_bwd_step")

    def convlstm16_supported(self, x, gates, pk, g, F):
        """16-bit ConvLSTM with the cell update in the recurrent convolution's epilogue (wdg_convlstm_step_h16)?"""
        plan, _, _ = self._plan(x, gates, pk.cin, pk.cout, g)
        return bool(self.lib.wdg_convlstm_h16_supported(plan, F))

    def convlstm16_gates(self, x, pk, bias, gates, g, F, fmt="bf16"):
        """Input part of the gates for all timesteps, gate columns interleaved (the layout the step kernel reads)."""
        plan, _, _ = self._plan(x, gates, pk.cin, pk.cout, g)
        if self._is16(x, fmt):           # the layer input in the operand format (written that way by its producer)
            native.check(self.lib.wdg_conv_fwd_h16_gates_x16(plan, x.data_ptr(), 1, pk.half(fmt)[0].data_ptr(), _ptr(bias), gates.data_ptr(),
                                                             F, 0 if fmt == "bf16" else 1, self.stream), "conv_fwd_h16_gates_x16")
            return
        native.check(self.lib.wdg_conv_fwd_h16_gates(plan, x.data_ptr(), pk.half(fmt)[0].data_ptr(), _ptr(bias), gates.data_ptr(),
                                                     F, 0 if fmt == "bf16" else 1, self.stream), "conv_fwd_h16_gates")

    def convlstm16_step(self, h_prev, pk, gates_t, c_prev, c_out, h_out, g, F, fmt="bf16"):
        """h_t, c_t from h_{t-1}, c_{t-1} (None at t = 0) and the input part of the gates of timestep t."""
        plan, _, _ = self._plan(h_out if h_prev is None else h_prev, gates_t, pk.cin, pk.cout, g)
        _, ldc, _ = _v4(c_out)
        _, ldh, _ = _v4(h_out)
        if self._is16(h_out, fmt):
            # the hidden state ONLY in the operand format: its readers in inference (the next step, the next convolution) round it
            # to that format while staging — the same bits, half the bytes read, no fp32 copy written
            assert h_prev is None or self._is16(h_prev, fmt)
            native.check(self.lib.wdg_convlstm_step_h16x(plan, _ptr(h_prev), 1, pk.half(fmt)[0].data_ptr(), gates_t.data_ptr(), _ptr(c_prev),
                                                         c_out.data_ptr(), ldc, None, 0, h_out.data_ptr(), ldh, F, 0 if fmt == "bf16" else 1,
                                                         self.stream), "convlstm_step_h16x")
            return
        native.check(self.lib.wdg_convlstm_step_h16(plan, _ptr(h_prev), pk.half(fmt)[0].data_ptr(), gates_t.data_ptr(), _ptr(c_prev),
                                                    c_out.data_ptr(), ldc, h_out.data_ptr(), ldh, F, 0 if fmt == "bf16" else 1,
                                                    self.stream), "convlstm_step_h16")

    def conv_dgrad_bf16(self, dy, pk, dx, g, bias=None, act=False, affine=None, accumulate=False, slope=0.2, fmt="bf16"):
        plan, _, _ = self._plan(dx, dy, pk.cin, pk.cout, g)
        in16, out16 = self._is16(dy, fmt), self._is16(dx, fmt)
        if in16 or out16:
            assert not accumulate
            native.check(self.lib.wdg_conv_dgrad_h16_act16(plan, dy.data_ptr(), in16, pk.half(fmt)[1].data_ptr(), 0 if fmt == "bf16" else 1,
                                                           _ptr(bias), _ptr(affine), dx.data_ptr(), out16, int(act), slope, self.stream),
                         "conv_dgrad_h16_act16")
            return
        fn = self.lib.wdg_conv_dgrad_bf16 if fmt == "bf16" else self.lib.wdg_conv_dgrad_f16
        native.check(fn(plan, dy.data_ptr(), pk.half(fmt)[1].data_ptr(), _ptr(bias), _ptr(affine),
                        dx.data_ptr(), int(act), slope, int(accumulate), self.stream), "conv_dgrad_16")

    H16_DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16}

    def act16_output_conv_ok(self, pk_up, pk_out, g_out, x_low=None, bias=None, affine=None):
        """True when the generator's last two layers can pass their 16-channel activation in the 16-bit operand format: the fused
        upsample kernel writes it (out16), the 16 -> (<= 4) output conv reads it (wdg_conv_thin16_fwd_h16).  Same values as the
        fp32 hand-over (the reader rounds to the operand format either way), half the bytes."""
        if x_low is not None and (x_low.shape[3] != pk_up.cout or _v4(x_low)[1] % (8 if x_low.dtype != torch.float32 else 4)):
            return False
        if any(t is not None and t.data_ptr() % 16 for t in (bias, affine)):
            return False
        return bool(self.act16 and self._upconv_fused16_layer_ok(pk_up)
                    and g_out.kh == 3 and g_out.kw == 3 and g_out.stride == 1 and g_out.pad == 1 and pk_out.cin == 16 and pk_out.cout <= 4)

    def _upconv_fused16_layer_ok(self, pk):
        """The layer-level preconditions of the ONE route of upconv_fwd_bf16 that accepts activations in the 16-bit operand
        format (the fused kernel, csrc/upconv_fused_h16.hip).  Shared by the act16_* predicates the networks use to ALLOCATE
        16-bit buffers and by upconv_fwd_bf16 itself, so that a switch such as WDG_UPCONV_COLFWD=0 cannot leave a 16-bit buffer
        in front of a route that would treat it as fp32."""
        return bool(self.upconv_colfwd and self.upconv_fused16 and not self.z16 and pk.w is not None and pk.cout % 8 == 0 and
                    pk.wD is pk.w and self.lib.wdg_upconv_col_supported(pk.cin) and
                    self.lib.wdg_upconv_fused_h16_supported(pk.cout, pk.cin))

    def act16_upconv_in_ok(self, x_low, pk_up):
        """The fused upsample kernel reads x_low in the 16-bit operand format."""
        return bool(self.act16 and self._upconv_fused16_layer_ok(pk_up) and x_low.shape[3] == pk_up.cout and _v4(x_low)[1] % 8 == 0)

    def act16_lstm_ok(self, x, gates, pk_x, pk_h, g, F):
        """Would the 16-bit ConvLSTM (convlstm16_gates / convlstm16_step) take its input AND keep its hidden state in the operand
        format?  (The state is written by the 128-column tile's transposed epilogue: whole tiles, 16-byte rows.)"""
        return bool(self.act16 and (4 * F) % 128 == 0 and _v4(x)[1] % 8 == 0 and F % 8 == 0 and
                    self.convlstm16_supported(x[:1], gates[:1], pk_h, g, F) and self.convlstm16_supported(x, gates, pk_x, g, F))

    def act16_conv_ok(self, x, y, pk, g, transposed, in16, out16):
        """Would conv_fwd_bf16 (transposed False) / conv_dgrad_bf16 (True) take x / y in the 16-bit operand format?  x, y: tensors
        of the final shapes and strides (any dtype: only the geometry is read)."""
        if not self.act16:
            return False
        plan, _, _ = self._plan(y, x, pk.cin, pk.cout, g) if transposed else self._plan(x, y, pk.cin, pk.cout, g)
        return bool(self.lib.wdg_conv_h16_act16_supported(plan, int(transposed), int(in16), int(out16)))

    def conv_halo_fwd_bf16(self, x, pk, bias, y, g, act=False, affine=None, slope=0.2, fmt="bf16"):
        """16-bit thin stride-1 conv (<= 64 output channels) through the halo-tile kernel.  x may be a tensor in the 16-bit operand
        format (3 x 3, 16 -> (<= 4) channels only: csrc/conv_halo_bf16.hip, wdg_conv_thin16_fwd_h16)."""
        if x.dtype in (torch.bfloat16, torch.float16):
            if x.dtype != self.H16_DTYPES[fmt]:
                raise ValueError(f"conv_halo_fwd_bf16: x is {x.dtype}, the operand format is {fmt}")
            n, H, W, _ = x.shape
            cp = (pk.cin + 3) // 4 * 4
            plan, _, _ = self._plan_dims(n, H, W, pk.cin, cp, H * W * cp, H, W, pk.cout, *_v4(y)[1:], g)
            px, ldx, isx = _v4(x)
            native.check(self.lib.wdg_conv_thin16_fwd_h16(plan, px, ldx, isx, pk.half(fmt)[0].data_ptr(), 0 if fmt == "bf16" else 1,
                                                          _ptr(bias), _ptr(affine), y.data_ptr(), int(act), slope, self.stream),
                         "conv_thin16_fwd_h16")
            return
        plan, _, _ = self._plan(x, y, pk.cin, pk.cout, g)
        fn = self.lib.wdg_conv_halo_fwd_bf16 if fmt == "bf16" else self.lib.wdg_conv_halo_fwd_f16
        native.check(fn(plan, x.data_ptr(), pk.half(fmt)[0].data_ptr(), _ptr(bias), _ptr(affine),
                        y.data_ptr(), int(act), slope, self.stream), "conv_halo_fwd_16")

    def upconv_fwd_bf16(self, x_low, pk, bias, y, g, act=True, affine=None, slope=0.2, fmt="bf16", pool=None):
        px, ldl, isl = _v4(x_low)
        py, ldy, isy = _v4(y)
        n, H, W, _ = y.shape
        y16 = y.dtype in (torch.bfloat16, torch.float16)
        x16 = self._is16(x_low, fmt)
        col = self.upconv_colfwd and g.kh == 5 and g.kw == 5 and g.stride == 1 and g.pad == 2 and pk.w is not None and \
            x_low.shape[3] == pk.cout and pk.cout % 8 == 0 and self.lib.wdg_upconv_col_supported(pk.cin) and \
            pk.wD is pk.w and (bias is None or bias.data_ptr() % 16 == 0) and (affine is None or affine.data_ptr() % 16 == 0)
        fused = col and self._upconv_fused16_layer_ok(pk) and ldl % (8 if x16 else 4) == 0
        if (y16 or x16) and not fused:
            # every other route below reads x_low / writes y as fp32: refuse BEFORE anything is launched
            raise ValueError("upconv_fwd_bf16: activations in the 16-bit operand format need the fused kernel "
                             "(act16_output_conv_ok / act16_upconv_in_ok), which this layer / switch setting does not take")
        if col:
            # column form with 16-bit GEMM operands: z = x * W, then the fp32 bilinear gather.  z16 (WDG_Z16=1, opt-in): z itself is
            # stored in the operand format — 400 columns per low-resolution pixel make it the largest tensor of the forward (1.4 GB
            # per 16-tile group in fp32), written once and read once.  Measured SLOWER than fp32 z twice: with 8-byte accesses of
            # four values (16-tile group 3.78 -> 4.23 ms) and with 16-byte accesses of eight (lane-pair exchange in the GEMM's
            # epilogue, eight-value slots in the gather: 3.82 -> 3.97 ms) — neither kernel is bound by z's bytes: the 400-column
            # GEMM has a reduction of only 160 (five MFMA K-steps per 16 stores) and the gather is bound by its LDS passes
            if fused:
                # both stages in one launch, z never leaves the CU (csrc/upconv_fused_h16.hip): 0.48 + 0.54 ms -> one launch per
                # 16-tile group of the shipped generator.  y in the operand format: for a reader that rounds to it anyway
                if y16 and y.dtype != self.H16_DTYPES[fmt]:
                    raise ValueError(f"upconv_fwd_bf16: y is {y.dtype}, the operand format is {fmt}")
                native.check(self.lib.wdg_upconv_fused_h16(px, ldl, isl, pk.half(fmt)[1].data_ptr(), 0 if fmt == "bf16" else 1, _ptr(bias),
                                                           _ptr(affine), py, ldy, isy, n, H // 2, W // 2, pk.cout, pk.cin, int(act), slope,
                                                           int(y16), x16, self.stream), "upconv_fused_h16")
                return
            plan16 = None
            if self.z16 and pk.cin % 8 == 0:
                plan16, _, _ = self._plan_dims(n, H // 2, W // 2, 25 * pk.cin, 25 * pk.cin, (H // 2) * (W // 2) * 25 * pk.cin,
                                               H // 2, W // 2, pk.cout, *_v4(x_low)[1:], ConvGeom(1, 1, 1, 0))
                if not self.lib.wdg_upconv_colgemm_h16_supported(plan16):
                    plan16 = None           # map outside the patch kernel's tile shapes: fp32 z route below
            if plan16 is not None:
                plan = plan16
                pool_ = self._scratch_bufs if pool is None else pool
                key, tdt = "upc_col16_" + fmt, (torch.bfloat16 if fmt == "bf16" else torch.float16)
                z16 = pool_.get(key)
                if z16 is None or tuple(z16.shape) != (n, H // 2, W // 2, 25 * pk.cin):
                    z16 = pool_[key] = torch.empty(n, H // 2, W // 2, 25 * pk.cin, dtype=tdt, device=self.device)
                f = 0 if fmt == "bf16" else 1
                native.check(self.lib.wdg_upconv_colgemm_h16(plan, x_low.data_ptr(), pk.half(fmt)[1].data_ptr(), z16.data_ptr(), f,
                                                             self.stream), "upconv_colgemm_h16")
                native.check(self.lib.wdg_upconv_gather_h16(z16.data_ptr(), f, _ptr(bias), _ptr(affine), py, ldy, isy, n, H // 2, W // 2,
                                                            pk.cin, int(act), slope, self.stream), "upconv_gather_h16")
                return
            z = self._scratch("upc_col", n, H // 2, W // 2, 25 * pk.cin, pool=pool)
            plan, _, _ = self._plan(z, x_low, 25 * pk.cin, pk.cout, ConvGeom(1, 1, 1, 0))
            fn = self.lib.wdg_conv_dgrad_bf16 if fmt == "bf16" else self.lib.wdg_conv_dgrad_f16
            native.check(fn(plan, x_low.data_ptr(), pk.half(fmt)[1].data_ptr(), 0, 0, z.data_ptr(), 0, slope, 0, self.stream),
                         "conv_dgrad_16")
            native.check(self.lib.wdg_upconv_gather(z.data_ptr(), _ptr(bias), _ptr(affine), py, ldy, isy, n, H // 2, W // 2,
                                                    pk.cin, int(act), slope, 0, 0, self.stream), "upconv_gather")
            return
        cp = (pk.cout + 3) // 4 * 4
        plan, _, _ = self._plan_dims(n, H, W, pk.cin, ldy, isy, H, W, pk.cout, cp, H * W * cp, g)
        fn = self.lib.wdg_upconv_fwd_bf16 if fmt == "bf16" else self.lib.wdg_upconv_fwd_f16
        native.check(fn(plan, px, ldl, isl, pk.half(fmt)[1].data_ptr(), _ptr(bias), _ptr(affine),
                        py, int(act), slope, self.stream), "upconv_fwd_16")

    def upconv_fwd(self, x_low, pk, bias, y, g, act=True, slope=0.2, pool=None, bn_stats=None, bn_affine=None):
        """y = act(convT(bilinear_x2(x_low), W) + bias) without materialising the upsampled tensor.
        pk/g describe the transposed conv as the conv it is the adjoint of (cin = y channels).
        bn_stats / bn_affine: BatchNormalization hooks as in conv_fwd (fused in the column-form gather kernel; the other
        routes run the standalone pass behind the convolution)."""
        if bn_stats is not None or bn_affine is not None:
            col = self.upconv_colfwd and g.kh == 5 and g.kw == 5 and g.stride == 1 and g.pad == 2 and pk.w is not None and \
                x_low.shape[3] == pk.cout and pk.cout % 4 == 0 and self.lib.wdg_upconv_col_supported(pk.cin) and \
                pk.wD is pk.w and (bias is None or bias.data_ptr() % 16 == 0) and (bn_affine is None or bn_affine.data_ptr() % 16 == 0)
            if not col:
                self.upconv_fwd(x_low, pk, bias, y, g, act=act, slope=slope, pool=pool)
                y2 = y.view(-1, y.shape[-1])[:, :(pk.cin + 3) // 4 * 4]      # (zero pad channels included: the norm's padded width)
                if bn_stats is not None:
                    self.bn_stats(y2, bn_stats[0])
                else:
                    self.bn_apply(y2, bn_affine, y2)
                return
        px, ldl, isl = _v4(x_low)
        py, ldy, isy = _v4(y)
        n, H, W, _ = y.shape
        if self.upconv_colfwd and g.kh == 5 and g.kw == 5 and g.stride == 1 and g.pad == 2 and pk.w is not None and \
                x_low.shape[3] == pk.cout and pk.cout % 4 == 0 and self.lib.wdg_upconv_col_supported(pk.cin) and \
                pk.wD is pk.w and (bias is None or bias.data_ptr() % 16 == 0):
            # column form: z = x * W on the low-res grid (1x1 GEMM, 25*cin columns), then the bilinear gather
            z = self._scratch("upc_col", n, H // 2, W // 2, 25 * pk.cin, pool=pool)
            self.conv_dgrad(x_low, pk.as_1x1(), z, ConvGeom(1, 1, 1, 0))
            native.check(self.lib.wdg_upconv_gather(z.data_ptr(), _ptr(bias), _ptr(bn_affine), py, ldy, isy, n, H // 2, W // 2, pk.cin,
                                                    int(act), slope, _ptr(bn_stats), bn_stats.shape[0] if bn_stats is not None else 0,
                                                    self.stream), "upconv_gather")
            return
        if self.upconv4 and g.kh == 5 and g.kw == 5 and g.stride == 1 and g.pad == 2 and pk.w is not None and \
                self.lib.wdg_upconv4_supported(pk.cin, pk.cout, H // 2, W // 2):
            # four composite 4x4 convolutions on the low-res grid (16 instead of 25 taps per output pixel)
            native.check(self.lib.wdg_upconv4_fwd(px, ldl, isl, n, H // 2, W // 2, pk.cout, pk.up4().data_ptr(), _ptr(bias),
                                                  py, ldy, isy, pk.cin, int(act), slope, self.stream), "upconv4_fwd")
            return
        cp = (pk.cout + 3) // 4 * 4
        plan, _, _ = self._plan_dims(n, H, W, pk.cin, ldy, isy, H, W, pk.cout, cp, H * W * cp, g)
        native.check(self.lib.wdg_upconv_fwd(plan, px, ldl, isl, pk.wD.data_ptr(), _ptr(bias), py, int(act), slope,
                                             self.stream), "upconv_fwd")

    def _scratch(self, key, *shape, pool=None):
        """Scratch tensor of the caller's `pool` (a dict the caller owns).  Networks pass the dict that lives in their
        resident buffer set, next to the HIP graphs captured on those buffers: a captured graph bakes the scratch address
        in, so the scratch must live and die with the graphs, and a second live network (other image size / T) must not
        be able to free it.  Without a pool: the process-wide one (eager callers only)."""
        pool = self._scratch_bufs if pool is None else pool
        t = pool.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            t = pool[key] = self.empty(*shape)
        return t

    def upconv_bwd(self, x_low, dpre, pk, dw, dx_low, g, pool=None, wgrad_async=None):
        """Backward of y = convT(bilinear_x2(x_low), W) given dpre = dL/dy (before bias/activation):
        dw += dL/dW, dx_low = dL/dx_low.  pk/g as in upconv_fwd.  5x5 layers with 4/8/16 output channels run in
        column form on the low-res grid (csrc/upconv_col.hip: a quarter of the multiply-adds); anything else
        through the materialised upsampled tensor.  wgrad_async(fn): runs the weight-gradient launch off the caller's stream."""
        n, Hl, Wl, C = x_low.shape
        lib = self.lib
        if self.upconv_col and g.kh == 5 and g.kw == 5 and g.stride == 1 and g.pad == 2 and pk.w is not None and \
                C == pk.cout and pk.cout % 4 == 0 and lib.wdg_upconv_col_supported(pk.cin) and pk.wD is pk.w:
            pdy, lddy, isdy = _v4(dpre)
            col = self._scratch("upc_col", n, Hl, Wl, 25 * pk.cin, pool=pool)
            native.check(lib.wdg_upconv_col(pdy, lddy, isdy, col.data_ptr(), n, Hl, Wl, pk.cin, self.stream), "upconv_col")
            pk1, g1 = pk.as_1x1(), ConvGeom(1, 1, 1, 0)
            wg = lambda: self.conv_wgrad(col, x_low, pk1, dw.view(1, 1, 25 * pk.cin, pk.cout), g1, accumulate=True)  # noqa: E731
            if wgrad_async is not None:
                wgrad_async(wg)            # (the caller's pass joins before the column scratch is written again)
            else:
                wg()
            self.conv_fwd(col, pk1, None, dx_low, g1, act=False)
            return
        up = self._scratch("upc_up", n, 2 * Hl, 2 * Wl, C, pool=pool)
        dup = self._scratch("upc_dup", n, 2 * Hl, 2 * Wl, C, pool=pool)
        self.upsample2x_fwd(x_low, up)
        self.conv_wgrad(dpre, up, pk, dw, g, accumulate=True)
        self.conv_fwd(dpre, pk, None, dup, g, act=False)
        self.upsample2x_bwd(dup, dx_low)

    def conv_wgrad(self, x, dy, pk, dw, g, accumulate=True, dbias=None):
        """dw[kh,kw,Cin,Cout] (+)= x (*) dy;  dbias[Cout] += sum_pixels dy when given."""
        plan, wsb, _ = self._plan(x, dy, pk.cin, pk.cout, g)
        ws = self._workspace(wsb)
        assert dw.is_contiguous()
        if dbias is not None:
            native.check(self.lib.wdg_conv_wgrad_bias(plan, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), dbias.data_ptr(),
                                                      int(accumulate), ws.data_ptr(), ws.numel(), self.stream),
                         "conv_wgrad_bias")
            return
        native.check(self.lib.wdg_conv_wgrad(plan, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), int(accumulate),
                                             ws.data_ptr(), ws.numel(), self.stream), "conv_wgrad")

    def sn_power_iter(self, w2d, u):
        rows, cols = w2d.shape
        assert w2d.is_contiguous() and u.numel() == cols
        need = int(self.lib.wdg_sn_scratch_floats(rows, cols))
        if self._sn_scratch is None or self._sn_scratch.numel() < need:
            self._sn_scratch = self.empty(max(need, 1 << 16))
        native.check(self.lib.wdg_sn_power_iter(w2d.data_ptr(), u.data_ptr(), rows, cols,
                                                self._sn_scratch.data_ptr(), self.stream), "sn_power_iter")

    def make_prep_batch(self, entries):
        """entries: [(PackedWeights, u or None)] of one network -> object with run(sn, pack_all): all spectral-norm
        power iterations and all weight repacks of the network in one launch per stage."""
        return _PrepBatch(self, entries)

    # ---- batch norm -------------------------------------------------------------------------
    def bn_stats(self, x, stats):
        px, ld = _v2(x)
        native.check(self.lib.wdg_bn_stats(px, x.shape[0], x.shape[1], ld, stats.data_ptr(), self.stream), "bn_stats")

    def bn_finalize_train(self, stats, count, gamma, beta, mmean, mvar, momentum, eps, ss, saved):
        """stats: [R, 2C] fp64 replica slabs (summed by the kernel)."""
        assert stats.dim() == 2 and stats.is_contiguous()
        native.check(self.lib.wdg_bn_finalize_train(stats.data_ptr(), stats.shape[0], float(count), gamma.data_ptr(), beta.data_ptr(),
                                                    mmean.data_ptr(), mvar.data_ptr(), momentum, eps, ss.data_ptr(),
                                                    saved.data_ptr(), gamma.numel(), self.stream), "bn_finalize_train")

    def bn_collapse(self, stats):
        """stats[0] = stats.sum(0) in place ([R, 2C] fp64)."""
        native.check(self.lib.wdg_bn_collapse(stats.data_ptr(), stats.shape[0], stats.shape[1] // 2, self.stream), "bn_collapse")

    def bn_finalize_infer(self, gamma, beta, mmean, mvar, eps, ss):
        native.check(self.lib.wdg_bn_finalize_infer(gamma.data_ptr(), beta.data_ptr(), mmean.data_ptr(),
                                                    mvar.data_ptr(), eps, ss.data_ptr(), gamma.numel(),
                                                    self.stream), "bn_finalize_infer")

    def bn_apply(self, x, ss, z):
        px, ldx = _v2(x)
        pz, ldz = _v2(z)
        native.check(self.lib.wdg_bn_apply(px, ldx, ss.data_ptr(), pz, ldz, x.shape[0], x.shape[1], self.stream), "bn_apply")

    def bn_bwd_reduce(self, dz, y, saved, red):
        pdz, lddz = _v2(dz)
        py, ldy = _v2(y)
        native.check(self.lib.wdg_bn_bwd_reduce(pdz, lddz, py, ldy, saved.data_ptr(), dz.shape[0], dz.shape[1],
                                                red.data_ptr(), self.stream), "bn_bwd_reduce")

    def bn_bwd_apply(self, dz, y, saved, gamma, red_mean, red_param, count, act_slope, dpre, dgamma, dbeta, dbias):
        pdz, lddz = _v2(dz)
        py, ldy = _v2(y)
        pd, ldd = _v2(dpre)
        native.check(self.lib.wdg_bn_bwd_apply(pdz, lddz, py, ldy, saved.data_ptr(), gamma.data_ptr(),
                                               red_mean.data_ptr(), _ptr(red_param), float(count), act_slope,
                                               pd, ldd, _ptr(dgamma), _ptr(dbeta), _ptr(dbias), dz.shape[0],
                                               dz.shape[1], self.stream), "bn_bwd_apply")

    # ---- layer norm -------------------------------------------------------------------------
    def ln_fwd(self, y, gamma, beta, eps, z, mean_rstd):
        py, ldy = _v2(y)
        pz, ldz = _v2(z)
        native.check(self.lib.wdg_ln_fwd(py, ldy, gamma.data_ptr(), beta.data_ptr(), eps, pz, ldz, _ptr(mean_rstd),
                                         y.shape[0], y.shape[1], self.stream), "ln_fwd")

    def ln_bwd(self, dz, y, mean_rstd, gamma, act_slope, dpre, dgamma, dbeta, dbias):
        pdz, lddz = _v2(dz)
        py, ldy = _v2(y)
        pd, ldd = _v2(dpre)
        native.check(self.lib.wdg_ln_bwd(pdz, lddz, py, ldy, mean_rstd.data_ptr(), gamma.data_ptr(), act_slope, pd,
                                         ldd, _ptr(dgamma), _ptr(dbeta), _ptr(dbias), dz.shape[0], dz.shape[1],
                                         self.stream), "ln_bwd")

    # ---- ConvLSTM cell ----------------------------------------------------------------------
    def lstm_fwd(self, gates, c_prev, c, h, F):
        pg, ldg = _v2(gates)
        pcp, ldcp = _v2(c_prev) if c_prev is not None else (0, 0)
        pc, ldc = _v2(c)
        ph, ldh = _v2(h)
        native.check(self.lib.wdg_lstm_fwd(pg, ldg, pcp, ldcp, pc, ldc, ph, ldh, gates.shape[0], F, self.stream), "lstm_fwd")

    def lstm_bwd(self, gates, c_prev, c, dh, dc_in, dgates, dc_prev, F):
        pg, ldg = _v2(gates)
        pcp, ldcp = _v2(c_prev) if c_prev is not None else (0, 0)
        pc, ldc = _v2(c)
        pdh, lddh = _v2(dh)
        pdci, lddci = _v2(dc_in) if dc_in is not None else (0, 0)
        pdg, lddg = _v2(dgates)
        pdcp, lddcp = _v2(dc_prev) if dc_prev is not None else (0, 0)
        native.check(self.lib.wdg_lstm_bwd(pg, ldg, pcp, ldcp, pc, ldc, pdh, lddh, pdci, lddci, pdg, lddg, pdcp,
                                           lddcp, gates.shape[0], F, self.stream), "lstm_bwd")

    def convlstm1_supported(self, cin, F):
        return bool(self.lib.wdg_convlstm1_supported(cin, F))

    def convlstm1_x2_supported(self, cin, F, n2):
        return bool(self.lib.wdg_convlstm1_x2_supported(cin, F, n2))

    def convlstm1_fwd(self, x, wx, bias, h, cin, F, x2=None):
        """Fused single-timestep ConvLSTM: x [N,H,W,>=cin] -> h[..., :F] (gates are not stored).
        x2 = (tensor [N,H,W,>=n2], n2): the layer's LAST n2 input channels are read from that tensor instead of x (the
        high-res part of the [low | high] concatenation read in place — wdg_convlstm1_fwd_x2)."""
        px, ldx, isx = _v4(x)
        ph, ldh, ish = _v4(h)
        n, H, W, _ = x.shape
        if x2 is not None:
            p2, ld2, is2 = _v4(x2[0])
            native.check(self.lib.wdg_convlstm1_fwd_x2(px, ldx, isx, p2, ld2, is2, int(x2[1]), wx.data_ptr(), bias.data_ptr(), ph, ldh, ish,
                                                       n, H, W, cin, F, self.stream), "convlstm1_fwd_x2")
            return
        native.check(self.lib.wdg_convlstm1_fwd(px, ldx, isx, wx.data_ptr(), bias.data_ptr(), ph, ldh, ish, n, H, W,
                                                cin, F, self.stream), "convlstm1_fwd")

    def convlstm_gates_x_supported(self, x, gates, cin, F):
        return bool(self.gates_x and self.lib.wdg_convlstm_gates_x_supported(cin, F)) and gates.is_contiguous() and \
            gates.shape[3] == 4 * F and gates.data_ptr() % 16 == 0

    def convlstm_gates_x(self, x, wx, bias, gates, cin, F):
        """gates [N,H,W,4F] = conv(x, wx) + bias for all timesteps (the input part of the ConvLSTM pre-activations)."""
        px, ldx, isx = _v4(x)
        n, H, W, _ = x.shape
        native.check(self.lib.wdg_convlstm_gates_x(px, ldx, isx, wx.data_ptr(), bias.data_ptr(), gates.data_ptr(), n, H, W, cin, F,
                                                   self.stream), "convlstm_gates_x")

    def convlstm_gates_dx_supported(self, dgates, dx, cin, F):
        return bool(self.gates_x and self.lib.wdg_convlstm_gates_dx_supported(cin, F)) and dgates.is_contiguous() and \
            dgates.shape[3] == 4 * F and dgates.data_ptr() % 16 == 0

    def convlstm_gates_dx(self, dgates, wx, dx, cin, F, accumulate=False):
        """dx[..., :cin] (+)= conv_transpose(dgates, wx): the data gradient of convlstm_gates_x (2 -> 2-feature layer)."""
        pdx, lddx, isdx = _v4(dx)
        n, H, W, _ = dgates.shape
        native.check(self.lib.wdg_convlstm_gates_dx(dgates.data_ptr(), wx.data_ptr(), pdx, lddx, isdx, int(accumulate), n, H, W, cin, F,
                                                    self.stream), "convlstm_gates_dx")

    def convlstm1_dx_from_supported(self, cin, F, c0):
        return cin == 5 and F == 16 and c0 == 3

    def convlstm1_bwd(self, x, wx, bias, dh, dgates, dx, cin, F, accumulate_dx=False, dw=None, dbias=None, x2=None, dx_c0=0):
        """dgates [N,H,W,4F] (dense, optional) and dx[..., :cin] (optional) from x and dh, recomputing the gates.
        dw / dbias given: the kernel and bias gradient are accumulated in the same pass (no dgates tensor).
        dx_c0 > 0 (convlstm1_dx_from_supported): only the gradient of input channels [dx_c0, cin), written to dx[..., :cin - dx_c0]."""
        px, ldx, isx = _v4(x)
        pdh, lddh, isdh = _v4(dh)
        n, H, W, _ = x.shape
        pdx, lddx, isdx = _v4(dx) if dx is not None else (0, 0, 0)
        if dx_c0:
            assert dx is not None and dw is None and dgates is None and x2 is None and self.convlstm1_dx_from_supported(cin, F, dx_c0)
            native.check(self.lib.wdg_convlstm1_bwd_dx_from(px, ldx, isx, wx.data_ptr(), bias.data_ptr(), pdh, lddh, isdh, pdx, lddx, isdx,
                                                            int(accumulate_dx), n, H, W, cin, F, int(dx_c0), self.stream), "convlstm1_bwd_dx_from")
            return
        if x2 is not None:
            # (x2 as in convlstm1_fwd: the last channels of the input from a second tensor — wdg_convlstm1_bwd_x2)
            assert dgates is None and (dw is None or (dbias is not None and dw.is_contiguous()))
            p2, ld2, is2 = _v4(x2[0])
            ws = self._workspace(int(self.lib.wdg_convlstm1_wgrad_ws_bytes(n, H, W, cin, F))) if dw is not None else None
            native.check(self.lib.wdg_convlstm1_bwd_x2(px, ldx, isx, p2, ld2, is2, int(x2[1]), wx.data_ptr(), bias.data_ptr(), pdh, lddh, isdh,
                                                       pdx, lddx, isdx, int(accumulate_dx), n, H, W, cin, F, _ptr(dw), _ptr(dbias),
                                                       _ptr(ws), ws.numel() if ws is not None else 0, self.stream), "convlstm1_bwd_x2")
            return
        if dw is not None:
            assert dgates is None and dbias is not None and dw.is_contiguous()
            ws = self._workspace(int(self.lib.wdg_convlstm1_wgrad_ws_bytes(n, H, W, cin, F)))
            native.check(self.lib.wdg_convlstm1_bwd_wgrad(px, ldx, isx, wx.data_ptr(), bias.data_ptr(), pdh, lddh, isdh,
                                                          pdx, lddx, isdx, int(accumulate_dx), n, H, W, cin, F,
                                                          dw.data_ptr(), dbias.data_ptr(), ws.data_ptr(), ws.numel(),
                                                          self.stream), "convlstm1_bwd_wgrad")
            return
        if dgates is not None:
            assert dgates.is_contiguous() and dgates.shape[-1] == 4 * F
        native.check(self.lib.wdg_convlstm1_bwd(px, ldx, isx, wx.data_ptr(), bias.data_ptr(), pdh, lddh, isdh,
                                                _ptr(dgates), pdx, lddx, isdx, int(accumulate_dx), n, H, W, cin, F,
                                                self.stream), "convlstm1_bwd")

    def convln_supported(self, cin, cout):
        return bool(self.lib.wdg_convln_supported(cin, cout))

    def convln_fwd(self, x, w, bias, gamma, beta, eps, slope, y, z, mean_rstd):
        """y = lrelu(conv3x3(x, w) + bias), z = LN(y); w: master HWIO [3,3,cin,cout].  y = mean_rstd = None: only z
        is written (the backward is then convln_bwd_x, which recomputes them from x)."""
        px, ldx, isx = _v4(x)
        py, ldy, isy = _v4(y) if y is not None else (0, 0, 0)
        pz, ldz, isz = _v4(z)
        n, H, W, _ = z.shape
        native.check(self.lib.wdg_convln_fwd(px, ldx, isx, w.data_ptr(), bias.data_ptr(), gamma.data_ptr(),
                                             beta.data_ptr(), eps, slope, py, ldy, isy, pz, ldz, isz,
                                             _ptr(mean_rstd), n, H, W, w.shape[2], w.shape[3], self.stream), "convln_fwd")

    def convln_bwd_x(self, dz, x, w, bias, gamma, eps, slope, dx, dgamma, dbeta, dbias, dw):
        """Backward of convln_fwd from x: dx (optional), dgamma / dbeta / dbias += (all or none), dw += (optional)."""
        pdz, lddz, isdz = _v4(dz)
        px, ldx, isx = _v4(x)
        pdx, lddx, isdx = _v4(dx) if dx is not None else (0, 0, 0)
        n, H, W, _ = dz.shape
        ws = self._workspace(int(self.lib.wdg_convln_wgrad_ws_bytes(n, H, W, w.shape[2]))) if dw is not None else None
        native.check(self.lib.wdg_convln_bwd_x(pdz, lddz, isdz, px, ldx, isx, w.data_ptr(), bias.data_ptr(), gamma.data_ptr(),
                                               eps, slope, pdx, lddx, isdx, _ptr(dgamma), _ptr(dbeta), _ptr(dbias), _ptr(dw),
                                               _ptr(ws), ws.numel() if ws is not None else 0, n, H, W, w.shape[2],
                                               w.shape[3], self.stream), "convln_bwd_x")

    def convln_bwd(self, dz, y, mean_rstd, w, gamma, slope, dpre, dx, dgamma, dbeta, dbias):
        pdz, lddz, isdz = _v4(dz)
        py, ldy, isy = _v4(y)
        pdx, lddx, isdx = _v4(dx) if dx is not None else (0, 0, 0)
        n, H, W, _ = y.shape
        assert dpre.is_contiguous()
        native.check(self.lib.wdg_convln_bwd(pdz, lddz, isdz, py, ldy, isy, mean_rstd.data_ptr(), w.data_ptr(),
                                             gamma.data_ptr(), slope, dpre.data_ptr(), pdx, lddx, isdx, _ptr(dgamma),
                                             _ptr(dbeta), _ptr(dbias), n, H, W, w.shape[2], w.shape[3], self.stream), "convln_bwd")

    # ---- resampling / head ------------------------------------------------------------------
    def upsample2x_fwd(self, x, y):
        px, ldx, isx = _v4(x)
        py, ldy, isy = _v4(y)
        n, H, W, Cc = x.shape
        native.check(self.lib.wdg_upsample2x_fwd(px, ldx, isx, py, ldy, isy, n, H, W, Cc, self.stream), "upsample_fwd")

    def upsample2x_bwd(self, dy, dx, accumulate=False):
        pdy, lddy, isdy = _v4(dy)
        pdx, lddx, isdx = _v4(dx)
        n, H, W, Cc = dx.shape
        native.check(self.lib.wdg_upsample2x_bwd(pdy, lddy, isdy, pdx, lddx, isdx, n, H, W, Cc, int(accumulate),
                                                 self.stream), "upsample_bwd")

    def patch_gather(self, x, out, k, stride, pad):
        """out [n,t,t,k*k*C] = the t x t grid of k x k windows of x [n,H,W,C] (zero padded); see wdg_patch_gather."""
        px, ldx, isx = _v4(x)
        n, H, W, Cc = x.shape
        assert out.is_contiguous() and out.shape[0] == n and out.shape[3] == k * k * Cc
        native.check(self.lib.wdg_patch_gather(px, ldx, isx, out.data_ptr(), n, H, W, Cc, k, stride, pad, out.shape[1],
                                               self.stream), "patch_gather")

    def patch_scatter(self, dpatch, dx, k, stride, pad, accumulate=False):
        pdx, lddx, isdx = _v4(dx)
        n, H, W, Cc = dx.shape
        assert dpatch.is_contiguous() and dpatch.shape[3] == k * k * Cc
        native.check(self.lib.wdg_patch_scatter(dpatch.data_ptr(), pdx, lddx, isdx, n, H, W, Cc, k, stride, pad,
                                                dpatch.shape[1], int(accumulate), self.stream), "patch_scatter")

    def dense_gap_fwd(self, x, w, b, score, B, T):
        assert x.is_contiguous()
        native.check(self.lib.wdg_dense_gap_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), score.data_ptr(), B, T,
                                                x.shape[1], self.stream), "dense_gap_fwd")

    def dense_gap_bwd(self, x, w, dscore, dx, dw, db, B, T):
        native.check(self.lib.wdg_dense_gap_bwd(x.data_ptr(), w.data_ptr(), dscore.data_ptr(), _ptr(dx), _ptr(dw),
                                                _ptr(db), B, T, x.shape[1], self.stream), "dense_gap_bwd")

    def dense_gap_bwd_ln(self, x, w, dscore, dx, dw, db, B, T, y, mean_rstd, gamma, C, act_slope, dgamma, dbeta, dbias, par_ws=None):
        """dense_gap_bwd chained with the backward of the LayerNormalization (+ LeakyReLU) that produced x from y [rows * K / C, C]:
        dx receives the gradient w.r.t. that norm's producer's pre-activation (wdg_dense_gap_bwd_ln: dense_gap_bwd + ln_bwd in
        one launch, dz stays in registers)."""
        native.check(self.lib.wdg_dense_gap_bwd_ln(x.data_ptr(), w.data_ptr(), dscore.data_ptr(), dx.data_ptr(), _ptr(dw), _ptr(db), B, T,
                                                   x.shape[1], y.data_ptr(), mean_rstd.data_ptr(), gamma.data_ptr(), int(C), float(act_slope),
                                                   _ptr(dgamma), _ptr(dbeta), _ptr(dbias), _ptr(par_ws), self.stream), "dense_gap_bwd_ln")

    # ---- elementwise / reductions -----------------------------------------------------------
    def copy_channels(self, src, dst, accumulate=False):
        """dst[..., :C] (+)= src[..., :C] for 4-D views (N,H,W,C) (any channel alignment)."""
        n, H, W, Cc = src.shape
        assert dst.shape == src.shape

        def strides(t):
            ld = t.stride(2) if W > 1 else (t.stride(1) if H > 1 else max(Cc, t.stride(2)))
            if H > 1:
                assert t.stride(1) == W * ld
            return ld, (t.stride(0) if n > 1 else H * W * ld)
        lds, iss = strides(src)
        ldd, isd = strides(dst)
        native.check(self.lib.wdg_copy_channels(src.data_ptr(), lds, iss, dst.data_ptr(), ldd, isd, n, H * W, Cc,
                                                int(accumulate), self.stream), "copy_channels")

    def permute_bt(self, src, dst):
        """dst[t, b, ..., :C] = src[b, t, ..., :C] for 5-D views (outer, inner, H, W, C) -> (inner, outer, H, W, >=C)
        (or the reverse, whichever the shapes say): one launch for the API <-> time-major permutation."""
        no, ni, H, W, Cc = src.shape
        assert tuple(dst.shape[:4]) == (ni, no, H, W)
        C = min(Cc, dst.shape[4])
        lds, ldd = (src.stride(3) if W > 1 else src.stride(2)), (dst.stride(3) if W > 1 else dst.stride(2))
        if H > 1 and W > 1:
            assert src.stride(2) == W * lds and dst.stride(2) == W * ldd
        native.check(self.lib.wdg_copy_channels_2level(src.data_ptr(), lds, src.stride(1), src.stride(0), dst.data_ptr(), ldd,
                                                       dst.stride(0), dst.stride(1), no, ni, H * W, C, 0, self.stream),
                     "copy_channels_2level")

    def colsum(self, x, out, accumulate=True):
        px, ld = _v2(x)
        native.check(self.lib.wdg_colsum(px, ld, x.shape[0], x.shape[1], out.data_ptr(), int(accumulate), self.stream), "colsum")

    def lerp_batch(self, a, b, eps, out, pixels_per_img, B):
        pa, lda = _v2(a)
        pb, ldb = _v2(b)
        po, ldo = _v2(out)
        native.check(self.lib.wdg_lerp_batch(pa, lda, pb, ldb, eps.data_ptr(), po, ldo, a.shape[0], pixels_per_img, B,
                                             a.shape[1], self.stream), "lerp_batch")

    def sumsq_batch_ch(self, x, pixels_per_img, T, B, out):
        px, ld = _v2(x)
        native.check(self.lib.wdg_sumsq_batch_ch(px, ld, pixels_per_img, T, B, x.shape[1], out.data_ptr(), self.stream), "sumsq_batch_ch")

    def segment_meansq(self, flat, offsets, out):
        native.check(self.lib.wdg_segment_meansq(flat.data_ptr(), offsets.data_ptr(), out.numel(), out.data_ptr(),
                                                 self.stream), "segment_meansq")

    # ---- evaluation metrics (csrc/metrics.hip) --------------------------------------------------------
    def metrics_pointwise(self, real, fake):
        """[B, 6] fp64 per-sample sums of the pointwise metrics (see wdg_metrics_pointwise) for dense [B,T,H,W,2] winds."""
        assert real.shape == fake.shape and real.shape[-1] == 2 and real.is_contiguous() and fake.is_contiguous()
        B = real.shape[0]
        out = torch.empty(B, 6, dtype=torch.float64, device=self.device)
        native.check(self.lib.wdg_metrics_pointwise(real.data_ptr(), fake.data_ptr(), real[0].numel() // 2, B, out.data_ptr(),
                                                    self.stream), "metrics_pointwise")
        return out

    def lsd_sums(self, real, fake, eps):
        """([B] fp64 sums of (10 log10 power ratio)^2, bins per sample).  The 2-D real FFT is rocFFT's (through
        torch.fft) over the LAST TWO axes of the 5-D tensor, as tf.signal.rfft2d is applied in metrics.py:123-126."""
        B = real.shape[0]
        sr = torch.view_as_real(torch.fft.rfft2(real)).contiguous()
        sf = torch.view_as_real(torch.fft.rfft2(fake)).contiguous()
        n = sr[0].numel() // 2
        out = torch.empty(B, dtype=torch.float64, device=self.device)
        native.check(self.lib.wdg_lsd_reduce(sr.data_ptr(), sf.data_ptr(), n, B, eps, out.data_ptr(), self.stream), "lsd_reduce")
        return out, n

    def spatial_ks(self, real, fake, patch, points):
        """[H-patch+1, W-patch+1] fp64 image of the mean KS statistic (wdg_spatial_ks); points: 100 ascending floats."""
        assert real.shape == fake.shape and real.is_contiguous() and fake.is_contiguous()
        B, T, H, W, Cc = real.shape
        pts = torch.as_tensor(points, dtype=torch.float32).to(self.device)
        assert pts.numel() == 100
        scratch = torch.empty(int(self.lib.wdg_spatial_ks_scratch_bytes(B, T, H, W, Cc)), dtype=torch.uint8, device=self.device)
        out = torch.empty(H - patch + 1, W - patch + 1, dtype=torch.float64, device=self.device)
        native.check(self.lib.wdg_spatial_ks(real.data_ptr(), fake.data_ptr(), B, T, H, W, Cc, patch, pts.data_ptr(),
                                             scratch.data_ptr(), out.data_ptr(), self.stream), "spatial_ks")
        return out

    def philox_normal(self, out, seed, offset, std, add=None):
        po, ldo = _v2(out)
        pa, lda = _v2(add) if add is not None else (0, 0)
        native.check(self.lib.wdg_philox_normal(po, ldo, pa, lda, out.shape[0], out.shape[1], seed & (2**64 - 1),
                                                offset, std, self.stream), "philox_normal")

    def input_assemble_ok(self, ci, cn, ld):
        return bool(self.input_fused and self.lib.wdg_input_assemble_supported(ci, cn, ld))

    def input_assemble(self, image, rows_out, B, XY, cn, seed, offset, std):
        """rows_out [T' * B * XY, ld] (time-major rows of the generator's input buffer) <- [image | noise | 0]: image [B, T', XY.., CI]
        (any batch / time strides, pixels dense), the noise = philox_normal(rows_out[:, CI:CI + cn], seed, offset, std)'s stream."""
        assert image.dim() == 5 and image.stride(4) == 1 and image.stride(3) == image.shape[4] and image.stride(2) == image.shape[3] * image.shape[4]
        po, ldo = _v2(rows_out)
        if rows_out.dtype != torch.float32:      # rows in the 16-bit operand format of the inference-precision layers
            fmt = {v: k for k, v in self.H16_DTYPES.items()}[rows_out.dtype]
            native.check(self.lib.wdg_input_assemble_h16(image.data_ptr(), image.stride(0), image.stride(1), image.shape[4], po, ldo,
                                                         rows_out.shape[0], B, XY, cn, seed & (2**64 - 1), offset, std,
                                                         0 if fmt == "bf16" else 1, B, 0, self.stream), "input_assemble_h16")
            return
        native.check(self.lib.wdg_input_assemble(image.data_ptr(), image.stride(0), image.stride(1), image.shape[4], po, ldo, rows_out.shape[0],
                                                 B, XY, cn, seed & (2**64 - 1), offset, std, self.stream), "input_assemble")

    def input_assemble_slots(self, image, rows_all, B, XY, cn, seed, offset, std, Bo, b0):
        """input_assemble for batch slots [b0, b0 + B) of the Bo slots of rows_all [T' * Bo * XY, ld]: the values of the dense
        B-slot call, written to rows (t * Bo + b0 + b) * XY + r (wdg_input_assemble_slots)."""
        assert image.dim() == 5 and image.stride(4) == 1 and image.stride(3) == image.shape[4] and image.stride(2) == image.shape[3] * image.shape[4]
        po, ldo = _v2(rows_all)
        Tn = image.shape[1]
        assert rows_all.shape[0] == Tn * Bo * XY and image.shape[0] == B
        if rows_all.dtype != torch.float32:
            fmt = {v: k for k, v in self.H16_DTYPES.items()}[rows_all.dtype]
            native.check(self.lib.wdg_input_assemble_h16(image.data_ptr(), image.stride(0), image.stride(1), image.shape[4], po, ldo, Tn * B * XY,
                                                         B, XY, cn, seed & (2**64 - 1), offset, std, 0 if fmt == "bf16" else 1, int(Bo), int(b0),
                                                         self.stream), "input_assemble_h16")
            return
        native.check(self.lib.wdg_input_assemble_slots(image.data_ptr(), image.stride(0), image.stride(1), image.shape[4], po, ldo, Tn * B * XY,
                                                       B, XY, cn, seed & (2**64 - 1), offset, std, int(Bo), int(b0), self.stream),
                     "input_assemble_slots")

    def philox_uniform(self, out, seed, offset):
        native.check(self.lib.wdg_philox_uniform(out.data_ptr(), out.numel(), seed & (2**64 - 1), offset, self.stream), "philox_uniform")

    def zero_ranges(self, flat, ranges):
        """flat[a:b] = 0 for every (a, b) of `ranges` (multiples of 4 elements) in one launch per 16 ranges (wdg_zero_ranges)."""
        for k in range(0, len(ranges), 16):
            part = ranges[k:k + 16]
            arr = (C.c_int64 * (2 * len(part)))(*[int(x) for ab in part for x in ab])
            native.check(self.lib.wdg_zero_ranges(flat.data_ptr(), arr, len(part), self.stream), "zero_ranges")

    def adam_tf(self, p, g, m, v, lr_t, beta1, beta2, eps, grad_scale=1.0):
        native.check(self.lib.wdg_adam_tf(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr_t,
                                          beta1, beta2, eps, grad_scale, self.stream), "adam_tf")
