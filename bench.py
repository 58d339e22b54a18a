#!/usr/bin/env python3
"""Headline benchmark: full GAN train step (G + D + Adam, n_critic = 3, metrics recompute included —
exactly GAN.train_step of the reference, ganbase.py:21-94) on synthetic 2-channel 32x32 -> 256x256
fp32 wind tiles, batch 32 per GPU, T = 1 (BASELINE.json configs[1]; weak scaling over GPUs).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events around every launch of the
conv kernels on their launch stream inside the timed region; `cpu_baseline` times the CPU restatement
(oracle/torch_model.py, torch-CPU fp32 on all host cores) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "wind-downscaling-gan_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
S, T, CIN, NZ, CH = 256, 1, 3, 20, 2


def synthetic_batch(B, seed, device):
    """configs[1]: 32x32 N(0,1) winds regridded x8 (nearest) + N(0,1) DEM at 256x256; N(0,1) high-res winds."""
    rng = np.random.default_rng(seed)
    wind = rng.standard_normal((B, T, S // 8, S // 8, 2)).astype(np.float32)
    wind = np.repeat(np.repeat(wind, 8, axis=2), 8, axis=3)
    dem = rng.standard_normal((B, T, S, S, 1)).astype(np.float32)
    low = np.concatenate([wind, dem], -1)
    high = rng.standard_normal((B, T, S, S, CH)).astype(np.float32)
    return torch.from_numpy(low).to(device), torch.from_numpy(high).to(device)


def generator_flops(net, executed=False):
    """Forward FLOPs of make_generator per tile-timestep (2 x MACs of every convolution, models.py:32-71), derived from the
    layer objects of the network that runs — kernel size, channels and the map each layer writes.  executed=True prices the
    upsample + 5x5 transposed-conv block as the kernels run it (column form on the low-resolution grid: a 1x1 GEMM with
    25 * C_out columns per low-res pixel instead of 25 taps per high-res pixel)."""
    S, T = net.S, net.T
    conv = lambda c, px: 2.0 * c.g.kh * c.g.kw * c.cin * c.cout * px      # noqa: E731
    lstm = net.lstm
    f = conv(net.c0, (S // 2) ** 2) + conv(net.c2, (S // 4) ** 2) + conv(net.c5, (S // 4) ** 2)
    # input convolution of the gates; executed, T = 1: c_0 = 0, the forget gate is dead in all three directions -> 3 of 4 gates
    f += 2.0 * 9 * lstm.cin * (3 if executed and T == 1 else 4) * lstm.F * (S // 4) ** 2
    f += 2.0 * 9 * lstm.F * 4 * lstm.F * (S // 4) ** 2 * (T - 1) / T      # recurrent convolution (absent at t = 0)
    f += 2.0 * net.c7.cout * net.c7.cin * (S // 2) ** 2                   # 2x2 stride-2 transposed: one tap per output pixel
    c9 = net.c9
    f += 2.0 * c9.cout * 25 * c9.cin * (S // 2) ** 2 if executed else conv(c9, S * S)
    f += conv(net.c11, S * S)
    return f


def discriminator_flops(net, executed=False):
    """Forward FLOPs of make_discriminator per tile-timestep (models.py:93-138), from the layer objects.  executed=True: at
    T = 1 the fused ConvLSTM kernels (csrc/convlstm1.hip) skip the forget gate (c_0 = 0)."""
    S, T = net.S, net.T
    f = 0.0
    for l in (net.lstm_a, net.lstm_b):
        f += 2.0 * 9 * l.cin * (3 if executed and T == 1 else 4) * l.F * S * S + 2.0 * 9 * l.F * 4 * l.F * S * S * (T - 1) / T
    for c in (net.conv_a, net.conv_b):
        f += 2.0 * 9 * c.cin * c.cout * S * S
    for conv, _, osz, _ in net.blocks:
        f += 2.0 * conv.g.kh * conv.g.kw * conv.cin * conv.cout * osz * osz
    return f + 2.0 * net.K


class ConvTimer:
    """HIP-event timing of every conv launch on the stream it is launched on (torch's current stream)."""

    def __init__(self):
        self.records = []

    def wrap(self, ops):
        from downscaling.engine import hipops
        timer = self

        def flops(x, y, pk, g):
            n, Ho, Wo = y.shape[0], y.shape[1], y.shape[2]
            return 2.0 * n * Ho * Wo * pk.cout * g.kh * g.kw * pk.cin

        orig_fwd, orig_dgrad, orig_wgrad = ops.conv_fwd, ops.conv_dgrad, ops.conv_wgrad

        def timed(name, fl, fn, *a, layer=None, **k):
            if torch.cuda.is_current_stream_capturing():
                return fn(*a, **k)        # inside a HIP-graph capture (HipOps.chain: the per-timestep loops at T > 1): not timed
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(*a, **k)
            e1.record()
            timer.records.append((name, fl, e0, e1, layer))

        def geom(which, x, y, pk, g):
            return f"{which:5s} {g.kh}x{g.kw}s{g.stride} {pk.cin:3d}->{pk.cout:3d} in {tuple(x.shape[:3])} out {tuple(y.shape[:3])}"

        def conv_fwd(x, pk, bias, y, g, **k):
            timed(ops.conv_kernel_label("fwd", x, y, pk, g), flops(x, y, pk, g), orig_fwd, x, pk, bias, y, g,
                  layer=geom("fwd", x, y, pk, g), **k)

        def conv_dgrad(dy, pk, dx, g, **k):
            timed(ops.conv_kernel_label("dgrad", dx, dy, pk, g), flops(dx, dy, pk, g), orig_dgrad, dy, pk, dx, g,
                  layer=geom("dgrad", dx, dy, pk, g), **k)

        def conv_wgrad(x, dy, pk, dw, g, **k):
            timed(ops.conv_kernel_label("wgrad", x, dy, pk, g), flops(x, dy, pk, g), orig_wgrad, x, dy, pk, dw, g,
                  layer=geom("wgrad", x, dy, pk, g), **k)

        orig_dslice, orig_wslice = ops.conv_dgrad_slice, ops.conv_wgrad_slice

        def conv_dgrad_slice(dy, pk, n0, n1, dx, g, **k):
            # (a channel range of the layer: the live gate columns of the generator's ConvLSTM at T = 1 — the FLOPs of the range)
            fl = 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * (n1 - n0) * g.kh * g.kw * pk.cin
            timed(ops.conv_kernel_label("dgrad", dx, dy, pk, g, cout=n1 - n0, w_ld=pk.cout), fl, orig_dslice, dy, pk, n0, n1, dx, g,
                  layer=geom("dgrad", dx, dy, pk, g) + f" cols[{n0}:{n1}]", **k)

        def conv_wgrad_slice(x, dy, pk, n0, n1, dw, g, **k):
            fl = 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * (n1 - n0) * g.kh * g.kw * pk.cin
            timed(ops.conv_kernel_label("wgrad", x, dy, pk, g, cout=n1 - n0, w_ld=pk.cout), fl, orig_wslice, x, dy, pk, n0, n1, dw, g,
                  layer=geom("wgrad", x, dy, pk, g) + f" cols[{n0}:{n1}]", **k)

        ops.conv_dgrad_slice, ops.conv_wgrad_slice = conv_dgrad_slice, conv_wgrad_slice
        orig_dgrad_ln = ops.conv_dgrad_lnbwd

        def conv_dgrad_lnbwd(dy, pk, dx, g, *a, **k):
            # data gradient -> LayerNorm + LeakyReLU backward of the producer in one call (epilogues 5 / 6 of wdg_igemm_kernel):
            # the convolution's FLOPs, timed with its fused norm backward
            # (or wdg_dgrad_s3_kernel, the patch kernel of the 7x7 stride-3 32 -> 64 layer, with its weight repack and — with
            # parameter gradients — its finish kernel)
            route = ops.conv_dgrad_lnbwd_route(dy, pk, dx, g, a[0], a[3], a[4], a[6] is not None or a[7] is not None or a[8] is not None)
            name = "wdg_dgrad_s3_kernel" if route == 2 else ops.conv_kernel_label("dgrad", dx, dy, pk, g)
            timed(name + " +LNbwd", flops(dx, dy, pk, g), orig_dgrad_ln, dy, pk, dx, g, *a, layer=geom("dgrad+LNbwd", dx, dy, pk, g), **k)

        ops.conv_dgrad_lnbwd = conv_dgrad_lnbwd
        orig_fwd_ln = ops.conv_fwd_ln

        def conv_fwd_ln(x, pk, bias, y, z, g, *a, **k):
            # conv -> LeakyReLU -> LayerNorm in one call: the convolution's FLOPs, timed with its fused norm
            timed(ops.conv_kernel_label("fwd", x, y, pk, g), flops(x, y, pk, g), orig_fwd_ln, x, pk, bias, y, z, g, *a,
                  layer=geom("fwd+LN", x, y, pk, g), **k)

        ops.conv_fwd_ln = conv_fwd_ln
        orig_up = ops.upconv_fwd

        def upconv_fwd(x_low, pk, bias, y, g, **k):
            # algorithmic FLOPs of the reference layer (bilinear x2, then 25 taps); the composite-kernel path
            # (csrc/upconv4.hip) executes 16/25 of these MACs
            n, Ho, Wo = y.shape[0], y.shape[1], y.shape[2]
            timed("wdg_upconv4_kernel (fused upsample + 5x5 transposed conv; executes 0.64 of the algorithmic MACs)",
                  2.0 * n * Ho * Wo * pk.cin * g.kh * g.kw * pk.cout, orig_up, x_low, pk, bias, y, g, layer="upconv", **k)

        ops.conv_fwd, ops.conv_dgrad, ops.conv_wgrad = conv_fwd, conv_dgrad, conv_wgrad
        if not getattr(ops, "upconv_colfwd", False):
            # (in column form the block is a 1x1 GEMM, timed through conv_dgrad above with the FLOPs it executes, plus
            # an untimed gather pass)
            ops.upconv_fwd = upconv_fwd

    def summary(self):
        agg = {}
        for name, fl, e0, e1, _ in self.records:
            a = agg.setdefault(name, [0.0, 0.0, 0])
            a[0] += fl
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
        return agg

    def per_layer(self, steps):
        """{layer geometry: (kernel, calls per step, ms per call, TFLOP/s)} — `bench.py --per-layer` (stderr table)."""
        agg = {}
        for name, fl, e0, e1, layer in self.records:
            a = agg.setdefault((layer, name), [0.0, 0.0, 0])
            a[0] += fl
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
        rows = [(k[0], k[1], v[2] / steps, 1e3 * v[1] / v[2], v[0] / v[1] * 1e-12, 1e3 * v[1] / steps) for k, v in agg.items()]
        return sorted(rows, key=lambda r: -r[5])


def cpu_baseline(batch=4):
    """The CPU restatement of the same train step (torch-CPU fp32, all host cores) on a bounded sample."""
    from oracle import torch_model as TM
    from downscaling.engine.networks import DiscriminatorNet, GeneratorNet
    from oracle.torch_backend import TorchOps
    from tests.helpers import Draws
    # torch-CPU conv kernels stop scaling (and oversubscribe badly) far below the 256 hardware threads of the
    # GPU node for a batch-1 sample: use at most 32 threads and report that number as `cores`.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    ops = TorchOps(torch.float32)
    gen = GeneratorNet(ops, S, CIN, NZ, CH, T, seed=1)     # only used to initialise weights with the TF names/shapes
    disc = DiscriminatorNet(ops, S, S, CIN, CH, T, seed=2)
    gw = {v.name: v.value.clone() for v in gen.params.vars}
    dw = {v.name: v.value.clone() for v in disc.params.vars}
    del gen, disc
    rng = np.random.default_rng(0)
    low = torch.from_numpy(rng.standard_normal((batch, T, S, S, CIN)).astype(np.float32))
    high = torch.from_numpy(rng.standard_normal((batch, T, S, S, CH)).astype(np.float32))

    class D32(Draws):
        def _normal(self, C):
            return super()._normal(C).float()

        def eps(self):
            return super().eps().float()
    draws = D32(7, batch, T, S, NZ, CH, 0.1)
    og, od = TM.AdamTF(1e-4), TM.AdamTF(4e-4)
    nsteps = 3                                            # 1 warm-up + 3 timed steps: ~12 s of CPU work on the GPU node's host cores
    TM.train_step(gw, dw, low, high, draws, og, od)       # warm-up (thread pool, oneDNN primitive caches, allocator)
    per_step = []
    for _ in range(nsteps):
        t0 = time.perf_counter()
        TM.train_step(gw, dw, low, high, draws, og, od)
        per_step.append(time.perf_counter() - t0)
    dt = sum(per_step)
    med, best = sorted(per_step)[len(per_step) // 2], min(per_step)
    out = {"value": batch / med, "unit": "samples/s", "cores": cores, "kind": "port",
           "host_threads": os.cpu_count() or 1,
           "value_best": batch / best, "s_per_step_median": med, "s_per_step_min": best,
           "sample": f"1 warm-up + {nsteps} timed full GAN train steps (n_critic=3), batch {batch}, {S}x{S}, T={T}, torch-CPU fp32 "
                     f"restatement (oracle/torch_model.py; TensorFlow is not installable here), {dt:.1f} s timed; value = batch / "
                     f"median step, value_best = batch / fastest step (BASELINE.md section 3)"}
    # BASELINE configs[0] (the reference's own CPU-runnable case): generator-only forward of G(128, T=1) on one
    # 16x16 ERA5 patch (x8 nearest) + 128x128 DEM, batch 1, 3 warm-up + 10 timed iterations
    gen = GeneratorNet(ops, 128, CIN, NZ, CH, 1, seed=1)
    gw0 = {v.name: v.value.clone() for v in gen.params.vars}
    del gen
    wind = np.repeat(np.repeat(np.random.default_rng(0).standard_normal((1, 1, 16, 16, 2)), 8, axis=2), 8, axis=3)
    image = torch.from_numpy(np.concatenate([wind, np.random.default_rng(1).standard_normal((1, 1, 128, 128, 1))], -1).astype(np.float32))
    noise = torch.from_numpy((0.1 * np.random.default_rng(2).standard_normal((1, 1, 128, 128, NZ))).astype(np.float32))
    with torch.no_grad():
        for _ in range(3):
            TM.generator_forward(gw0, image, noise, False)
        each = []
        for _ in range(10):
            t0 = time.perf_counter()
            TM.generator_forward(gw0, image, noise, False)
            each.append(time.perf_counter() - t0)
    dt0, best0 = sorted(each)[len(each) // 2], min(each)
    # (BASELINE.md section 3 words the baseline as "all host cores".  torch-CPU's conv kernels collapse when oversubscribed: at the GPU
    # node's 256 hardware threads the full train step took 376 s (profiles/r04a_bench.json) against 2.5 s at 32 threads, and this
    # 5.6-GFLOP forward 9 s against 6 ms — an oversubscription artefact, not a baseline: it is no longer timed or printed.)
    out["configs0_generator_forward_128"] = {"ms_per_forward": 1e3 * dt0, "ms_per_forward_min": 1e3 * best0, "samples_per_s": 1.0 / dt0,
                                             "cores": cores, "kind": "port",
                                             "host_threads": os.cpu_count() or 1,
                                             "sample": "G(128,3,20,2,T=1) forward, batch 1, 3 warm-up + 10 timed (median; min beside it), "
                                                       "torch-CPU fp32 restatement"}
    return out


def csrc_hash():
    """sha256 over the kernel sources (sorted csrc/*.hip, *.h): ties profiles/pmc_traffic.json to the code it measured
    (the GPU box has no .git)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted((ROOT / "wind-downscaling-gan_amd" / "csrc").glob("*.h*")):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()


def generator_leg(generator, gan, dev, batch=64, warm=3, iters=10):
    """The north star's "generator conv stack at batch 64" figure inside the default bench line: inference-mode forward
    of G(256,3,20,2,T=1) on 64 synthetic tiles with fresh Philox noise each step (inputs resident in HBM), timed with
    HIP events over `iters` forwards.  tflops_algorithmic prices the reference layer's FLOPs (22.385 GFLOP per sample, SURVEY
    8d); frac_executed only the multiply-adds the kernels execute — the upsample + 5x5 transposed-conv block runs in column
    form on the low-resolution grid (2*160*400 FLOP per low-res pixel instead of 2*25*160*16 per output pixel)."""
    net = generator.net
    low, _ = synthetic_batch(batch, 77, dev)
    net.set_image(low)

    def step():
        gan.noise_generator.prng.normal_into(net.noise_view(batch), 0.1)
        return net.forward(batch, training=False)
    for _ in range(warm):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    gf_alg, gf_exec = generator_flops(net), generator_flops(net, executed=True)   # 22.385 / 14.88 GFLOP at S = 256, T = 1
    return {"ms": ms, "batch": batch, "samples_per_s": batch / ms * 1e3,
            "tflops_algorithmic": gf_alg * batch / ms * 1e-9,
            "tflops_executed": gf_exec * batch / ms * 1e-9, "frac_executed": gf_exec * batch / ms * 1e-9 / PEAK_F32_MFMA_TFLOPS,
            "peak_tflops": PEAK_F32_MFMA_TFLOPS, "iters": iters, "warmup": warm, "target_ms": 18.2,
            "note": "whole forward incl. noise generation, norms, ConvLSTM cell.  frac_executed prices the multiply-adds the kernels "
                    "execute (the upsample + 5x5 block runs in column form on the low-res grid: a quarter of that block's reference "
                    "MACs, exact by linearity); tflops_algorithmic prices the reference layer's FLOPs and is a rate, not a roofline "
                    "fraction.  BASELINE.json target: >= 0.50 of the fp32-MFMA roofline at batch 64 = <= 18.2 ms"}


PEAK_16BIT_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (the 2:1-sparsity figure is not a bound)
PEAK_HBM_GBPS = 8000.0            # HBM3E spec peak (about 6.3 TB/s is what a streaming copy reaches)


def generator_activation_elements(size):
    """Algorithmic activation elements one generator forward moves per tile-timestep at tile edge `size` (inference): every
    inter-layer tensor written once and read once per consumer (models.py:24-71) — image 3 read, noise 20 written + read,
    layer-0 output 128 @ (S/2)^2 written + read twice (next conv, res_2 skip), layer-2 output 128 @ (S/4)^2 the same (res_4),
    the ConvLSTM's input gates 512 @ (S/4)^2 written + read, its h 128 @ (S/4)^2 written + read twice (next step, next conv),
    64 @ (S/4)^2, 32 @ (S/2)^2, 16 @ S^2 written + read, the 2-channel output written: 309 elements per output pixel."""
    return 309 * size * size


def generator_activation_bytes(size, act16):
    """Bytes behind generator_activation_elements on the 16-bit path: every tensor is stored in fp32 except — with the
    activation hand-over in the operand format (WDG_ACT16) — the layer-0 output (written once, read twice: 96
    elements per output pixel), the 32 @ (S/2)^2 tensor (16) and the 16 @ S^2 tensor (32): 144 of the 309 elements at 2 bytes."""
    return (309 * 4 - (144 * 2 if act16 else 0)) * size * size


def other_config_legs(dev, ops):
    """Driver-timed figures for BASELINE configs[0], [3], [4] inside the default bench line (N = 1 only; one small network
    build + a few hundred ms of GPU time).  All inputs resident in HBM before the timed regions except the end-to-end
    configs[3] call, whose upload / download are part of what it reports (phases beside it).  The 16-bit figures are NOT at
    the north star's 1e-4: their asserted bounds are 3e-2 (bf16) / 4e-3 (fp16) relative to the fp64 oracle
    (tests/test_configs_gpu.py), by design."""
    import downscaling.api as api
    from downscaling.gan.models import make_generator
    out = {}

    def events(fn, warm, iters):
        for _ in range(warm):
            fn()
        es = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
        torch.cuda.synchronize()
        es[0].record()
        for i in range(iters):
            fn()
            es[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(es[i].elapsed_time(es[i + 1]) for i in range(iters))
        return ms[len(ms) // 2], ms[0]

    # ---- configs[0]: generator-only forward, one 16x16 ERA5 patch (x8 nearest) + 128x128 DEM -> 128x128 wind, fp32
    rng = np.random.default_rng(0)
    wind = np.repeat(np.repeat(rng.standard_normal((1, 1, 16, 16, 2)), 8, axis=2), 8, axis=3)
    image = torch.from_numpy(np.concatenate([wind, np.random.default_rng(1).standard_normal((1, 1, 128, 128, 1))], -1).astype(np.float32)).to(dev)
    noise = torch.from_numpy((0.1 * np.random.default_rng(2).standard_normal((1, 1, 128, 128, NZ))).astype(np.float32)).to(dev)
    g0 = make_generator(128, CIN, NZ, CH, 1)
    med, best = events(lambda: g0([image, noise], training=False), 4, 20)
    gf0 = generator_flops(g0.net)                            # 5.596 GFLOP (S = 128, T = 1)
    out["config0_fwd_128"] = {"ms": med, "ms_min": best, "tflops": gf0 / med * 1e-9, "frac_of_mfma_f32_peak": gf0 / med * 1e-9 / PEAK_F32_MFMA_TFLOPS,
                              "dtype": "f32", "note": "G(128,3,20,2,T=1), batch 1, inputs resident, HIP-graph replay of the inference forward; one "
                                                      "tile cannot fill 256 CUs: latency-, not roofline-bound (5.6 GFLOP)"}
    del g0
    # ---- configs[3]: the shipped network G(96, T=24); one predict() group of 16 tiles, bf16 operands
    network = api.get_network(allow_random_init=True, random_seed=5)
    gen = network.generator
    # (api.predict_array runs SEVERAL predict() groups of 16 tiles per forward pass, each with its own noise draw — the group of 16
    # is the reference's noise-draw unit, api.py:132-137, not the launch unit; ms below is per group of 16)
    from downscaling.data.data_generator import LazyGroupNoise
    gpl = api.groups_per_forward(4)                          # (4 of a field's groups in one forward pass)
    tiles = torch.randn(16 * gpl, api.SEQUENCE_LENGTH, api.IMG_SIZE, api.IMG_SIZE, 3, device=dev)
    gen.inference_precision = "bf16"
    ngen = network.noise_generator
    if gpl == 1:
        fwd = lambda: gen([tiles, ngen.lazy(bs=16, channels=api.NOISE_CHANNELS)])       # noqa: E731
    else:
        fwd = lambda: gen([tiles, LazyGroupNoise(ngen, gpl, 16, ngen.noise_shape, api.NOISE_CHANNELS, ngen.std)])   # noqa: E731
    med, best = events(fwd, 4, 10)
    med, best = med / gpl, best / gpl
    tts = 16 * api.SEQUENCE_LENGTH
    gf_tt = generator_flops(gen.net)                         # 3.799 GFLOP per tile-timestep (S = 96, T = 24)
    fl = tts * gf_tt
    act16 = gen.net.buffers(16).get("cat2_bf16") is not None and bool(getattr(ops, "act16", False))
    by = tts * generator_activation_bytes(api.IMG_SIZE, act16)
    out["config3_bf16_group16_T24"] = {
        "ms": med, "ms_min": best, "groups_per_forward": gpl, "tile_timesteps_per_s": tts / med * 1e3, "dtype": "bf16 operands, f32 accumulate",
        "roofline": {"bound": "mfma", "achieved": fl / med * 1e-9, "peak": PEAK_16BIT_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": fl / med * 1e-9 / PEAK_16BIT_MFMA_TFLOPS},
        "hbm": {"algorithmic_bytes": by, "activations_in_operand_format": act16, "achieved_gbps": by / med * 1e-6, "peak_gbps": PEAK_HBM_GBPS,
                "frac": by / med * 1e-6 / PEAK_HBM_GBPS,
                "note": "309 activation elements per output pixel and tile-timestep (bench.generator_activation_elements), 4 bytes "
                        "each except the 144 handed over in the 16-bit operand format (bench.generator_activation_bytes); weights "
                        "(7.2 MB) excluded"},
        "note": "16 tiles x 24 h through G(96,3,20,2,T=24): input assembly (image + noise drawn on the device), forward replayed from "
                "its HIP graph, output permutation; parity bound 3e-2 vs the fp64 oracle"}
    # end to end: 1200 x 1200 x 24 h field -> 225 tiles -> blended field (upload and download included)
    fields = np.random.default_rng(3).standard_normal((24, 1200, 1200, 3)).astype(np.float32)
    fields[..., 2] = fields[..., 2] * 500 + 1200
    import contextlib
    import io
    runs = []
    with contextlib.redirect_stdout(io.StringIO()):          # predict() prints the reference's progress lines
        api.predict_array(fields, overlap_factor=0.05, network=network)
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            api.predict_array(fields, overlap_factor=0.05, network=network)
            torch.cuda.synchronize()
            runs.append(time.perf_counter() - t0)
        phases = {}
        api.predict_array(fields, overlap_factor=0.05, network=network, timings=phases)
        # the same with a caller-owned page-locked result buffer (predict_array(out=...)): the download at the link's rate
        pinned = torch.empty(24, 1200, 1200, 2, dtype=torch.float32, pin_memory=True)
        api.predict_array(fields, overlap_factor=0.05, network=network, out=pinned)
        runs_p = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            api.predict_array(fields, overlap_factor=0.05, network=network, out=pinned)
            torch.cuda.synchronize()
            runs_p.append(time.perf_counter() - t0)
        del pinned
    out["config3_end_to_end_1200"] = {"seconds": sorted(runs)[1], "seconds_min": min(runs), "tiles": 225,
                                      "seconds_pinned_result": sorted(runs_p)[1],
                                      "tile_timesteps_per_s": 225 * 24 / sorted(runs)[1], "dtype": "bf16 operands, f32 accumulate",
                                      "phase_seconds": {k: round(v, 4) for k, v in phases.items() if k != "laps"},
                                      "note": "api.predict_array on a host fp32 field (24,1200,1200,3), overlap_factor 0.05: upload, tile gather + "
                                              "normalisation, 15 groups of 16 tiles, crop + mean blend, download"}
    del fields
    # ---- configs[4]: 64 noise realisations x 8 tiles, fp16 operands (one GPU runs all 64; N ranks take 64 / N each)
    gen.inference_precision = "fp32"
    tiles8 = tiles[:8].contiguous()
    tiles = tiles[:16]
    api.predict_ensemble(tiles8, 64, network=network, precision="fp16")      # warm-up at the timed draw count (same batch shapes / graphs)
    runs4 = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ens = api.predict_ensemble(tiles8, 64, network=network, precision="fp16")
        torch.cuda.synchronize()
        runs4.append(time.perf_counter() - t0)
    dt = sorted(runs4)[2]
    out["config4_fp16_64x8"] = {"seconds": dt, "seconds_all": [round(x, 5) for x in runs4], "timing": "median of 5 calls after one warm-up call at 64 draws",
                                "realisations_per_s": 64 * 8 / dt, "tile_timesteps_per_s": 64 * 8 * 24 / dt,
                                "tflops": 64 * 8 * 24 * gf_tt / dt * 1e-12, "frac_of_16bit_mfma_peak": 64 * 8 * 24 * gf_tt / dt * 1e-12 / PEAK_16BIT_MFMA_TFLOPS,
                                "dtype": "fp16 operands, f32 accumulate", "finite": bool(torch.isfinite(ens).all()),
                                "note": "api.predict_ensemble(8 tiles, 64 draws), member-keyed Philox streams; parity bound 4e-3 vs the fp64 oracle"}
    return out


def child_legs(headline_ms):
    """Two driver-timed legs that need a process of their own (this one is idle meanwhile; N = 1 only):

    dp_path_one_rank — the data-parallel step as an 8-GPU run executes it per GPU: the same bench at batch 32, 256 x 256 in a
    ONE-rank RCCL process group (WDG_DIST_ALWAYS=1: asynchronous gradient all-reduces with the deferred Adam, SyncBN's
    statistics all-reduces, the metric reduce — real RCCL collectives with identity results).  It must be the headline's
    schedule (generator beside discriminator, twin discriminator): ms_per_step next to the headline's.

    train_T24_S96_b8 — the train step at the shipped shape (api.py:22 SEQUENCE_LENGTH = 24, 96 x 96 tiles, batch 8): the
    only configuration whose training exercises the ConvLSTM recurrence (models.py:45,93,101)."""
    import subprocess
    out = {}

    def run(extra_args, extra_env, timeout=600):
        env = dict(os.environ)
        env.update(extra_env)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--no-cpu-baseline", "--no-serial-pass", "--no-generator-leg",
               "--no-config-legs", "--no-split-leg", "--no-child-legs"] + extra_args
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=str(ROOT))
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            raise RuntimeError(f"child bench rc={r.returncode}: {r.stderr[-400:]}")
        return json.loads(lines[-1])

    try:
        j = run(["--steps", "6", "--warmup", "2"], {"WDG_DIST_ALWAYS": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29571"})
        out["dp_path_one_rank"] = {
            "ms_per_step": j["ms_per_step"], "value": j["value"], "unit": j["unit"], "vs_headline_ms": j["ms_per_step"] / headline_ms,
            "backend": j.get("rccl", {}).get("backend"), "sync_bn": j.get("rccl", {}).get("sync_bn"),
            "parallelism": j["config"]["parallelism"], "steps": j["steps"], "warmup": j["warmup"],
            "allreduce_ms_one_rank": j.get("rccl", {}).get("allreduce_ms"),
            "note": "fresh process, one-rank RCCL group (WDG_DIST_ALWAYS=1), batch 32, 256 x 256: the per-GPU step of the "
                    "data-parallel run — same multi-stream schedule as the headline plus the exchange code path"}
    except Exception as exc:          # a leg must never cost the headline line
        out["dp_path_one_rank"] = {"error": repr(exc)}
    try:
        # the same with every collective replaced by a stand-in kernel that occupies the chip as the 8-rank collective would
        # (engine/trainer.py DistSync, WDG_DP_PROXY=1): what RCCL's own stream — a fifth busy queue — and 8-rank SyncBN latencies
        # would do to the four-queue schedule, measurable on one GPU
        j = run(["--steps", "6", "--warmup", "2"], {"WDG_DIST_ALWAYS": "1", "WDG_DP_PROXY": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29573"})
        base = out.get("dp_path_one_rank", {}).get("ms_per_step")
        out["dp_fifth_queue_proxy"] = {
            "ms_per_step": j["ms_per_step"], "dp_fifth_queue_proxy_ms": (j["ms_per_step"] - base) if base else None,
            "vs_headline_ms": j["ms_per_step"] / headline_ms, "proxy": j.get("rccl", {}).get("proxy"),
            "note": "one-rank group, collectives replaced by wdg_dp_proxy launches: each flat-gradient all-reduce = 16 workgroups on a "
                    "fifth stream moving 2 * 7/8 of the buffer and lasting its ring time at 150 GB/s per link; each SyncBN / metric "
                    "all-reduce = a 10 us wait on the calling stream.  dp_fifth_queue_proxy_ms = this step - dp_path_one_rank's"}
    except Exception as exc:
        out["dp_fifth_queue_proxy"] = {"error": repr(exc)}
    try:
        j = run(["--size", "96", "--timesteps", "24", "--batch", "8", "--steps", "5", "--warmup", "2"], {})
        out["train_T24_S96_b8"] = {
            "tile_timesteps_per_s": j["value"], "ms_per_step": j["ms_per_step"], "steps": j["steps"], "warmup": j["warmup"],
            "step_tflops_algorithmic": j.get("step_tflops_algorithmic"),
            "frac_of_mfma_f32_peak_algorithmic": j.get("step_frac_of_mfma_f32_peak"), "frac_executed": j.get("step_frac_executed"),
            "host_enqueue_ms_per_step": j.get("host_enqueue_ms_per_step"),
            "note": "GAN.train_step on G(96,3,20,2,T=24) + D(96,96,3,2,T=24), batch 8 (192 tile-timesteps per step), fresh process; "
                    "41.7 GFLOP algorithmic per tile-timestep (7 Gf + 28 Df)"}
    except Exception as exc:
        out["train_T24_S96_b8"] = {"error": repr(exc)}
    return out


def bench_generator_forward(args, generator, gan, low, world, rank, dev):
    """Generator-only forward at `--batch` tiles per GPU (inference mode, fresh Philox noise each step)."""
    import torch.distributed as dist
    net, B = generator.net, low.shape[0]
    net.set_image(low)

    def step():
        gan.noise_generator.prng.normal_into(net.noise_view(B), 0.1)
        return net.forward(B, training=False, precision=args.precision)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        dist.destroy_process_group()
    if rank == 0:
        gf = generator_flops(net)  # 22.385 GFLOP per sample at S=256, T=1 (SURVEY §8 d)
        tf = gf * B * args.steps / dt * 1e-12
        print(json.dumps({
            "metric": "generator forward samples/s, 32x32->256x256 wind tiles", "value": world * B * args.steps / dt,
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else f"{args.precision} operands / f32 accumulate", "data": "synthetic",
            "config": {"workload": f"make_generator(256,3,20,2,T=1) forward, inference mode, batch {B}/GPU", "per_gpu_batch": B},
            "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                         "note": "whole generator forward (all kernels), algorithmic FLOPs 22.385 GFLOP/sample"}}), flush=True)


def rccl_evidence(dist, gan, world, rank, dev):
    """What the collectives of this run spanned, printed with the line so a reader can check it: backend, world size, every
    rank's (device index, name, PCI bus id, uuid) gathered over the process group, the bytes of the two gradient exchanges
    per step, and one discriminator-gradient all-reduce timed in isolation after the timed region (HIP events on the
    stream the collective is enqueued from; median of 5)."""
    backend = dist.get_backend()
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "device": int(torch.cuda.current_device()), "name": props.name,
            "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")),
            "pid": os.getpid()}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    dg, gg = gan.discriminator.net.params.grads, gan.generator.net.params.grads
    buf = torch.empty_like(dg)
    on_device = backend == "nccl"
    target = buf if on_device else buf.cpu()
    times = []
    for _ in range(7):
        if on_device:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            dist.barrier()
            e0.record()
            dist.all_reduce(target)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        else:
            dist.barrier()
            t0 = time.perf_counter()
            dist.all_reduce(target)
            times.append(1e3 * (time.perf_counter() - t0))
    times = sorted(times[2:])
    ids = {(d["pci_bus_id"], d["uuid"], d["device"]) for d in everyone}
    nb = dg.numel() * 4
    return {"backend": backend + (" (RCCL)" if on_device else " (host-staged functional check, not a benchmark)"),
            "world_size": world, "devices": everyone, "distinct_devices": len(ids),
            "grad_allreduce_bytes_per_step": 3 * nb + gg.numel() * 4,
            "d_grad_allreduce_bytes": nb, "allreduce_ms": times[len(times) // 2], "allreduce_ms_min": times[0],
            "allreduce_busbw_gbps": (2.0 * (world - 1) / world * nb / (times[len(times) // 2] * 1e-3) * 1e-9) if world > 1 else None,
            "sync_bn": bool(getattr(gan.engine.gen, "sync", None) is not None),
            "proxy": dict(gan.engine.sync.proxy_log, ranks=gan.engine.sync.proxy_ranks, link_gbps=gan.engine.sync.proxy_link_gbps,
                          small_us=gan.engine.sync.proxy_small_us, blocks=gan.engine.sync.proxy_blocks)
            if getattr(gan.engine.sync, "_proxy", False) else None,
            "note": "3 discriminator (34.2 MB) + 1 generator (7.2 MB) flat-gradient all-reduces per step, started asynchronously and "
                    "overlapped with the next network's forward; SyncBN adds 10 tiny fp64 [2C] all-reduces per generator pass"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE configs[1]: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-layer", action="store_true", help="print a per-layer table of the timed conv launches to stderr")
    ap.add_argument("--no-generator-leg", action="store_true", help="skip the generator-forward-at-batch-64 leg of the default line")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the configs[0] / [3] / [4] legs of the default line")
    ap.add_argument("--no-serial-pass", action="store_true",
                    help="skip the extra single-stream train steps the per-kernel roofline figures are taken from (profiling runs: "
                         "the figures then come from the timed region, whatever its stream schedule)")
    ap.add_argument("--no-split-leg", action="store_true", help="(accepted and ignored: the bf16-slice mode was removed in round 6)")
    ap.add_argument("--precision", choices=["fp32", "bf16", "fp16"], default="fp32", help="gen_fwd only: inference precision")
    ap.add_argument("--workload", choices=["train", "gen_fwd"], default="train",
                    help="train: the headline GAN train step (default); gen_fwd: generator-only forward (inference "
                         "mode) at --batch tiles, the 'generator conv stack at batch 64' figure of BASELINE.json")
    ap.add_argument("--no-child-legs", action="store_true",
                    help="skip the two legs that run in fresh child processes (dp_path_one_rank, train_T24_S96_b8): profiling runs "
                         "— a profiler's preloaded library must not see this process start other GPU programs")
    ap.add_argument("--no-sync-bn", action="store_true", help="per-replica BatchNorm statistics instead of SyncBN")
    ap.add_argument("--size", type=int, default=256, help="train only: tile edge (headline 256; the shipped network uses 96)")
    ap.add_argument("--timesteps", type=int, default=1,
                    help="train only: sequence length T (headline 1; 24 = the shipped SEQUENCE_LENGTH, which exercises the "
                         "ConvLSTM recurrence; reported as tile-timesteps/s, SURVEY 8d)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT",
                    help="wdg_set_tuning(KEY, INT) before anything is planned (A/B evidence runs; repeatable)")
    args = ap.parse_args()
    global S, T
    S, T = args.size, args.timesteps
    headline = (S, T) == (256, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` launches its own ranks: torch.distributed.run as a CHILD process (never an exec — nothing in
        # this process has touched the GPU yet, and it never will), one rank per GPU; rank 0's JSON line is relayed, the
        # child's return code becomes this process's.
        import subprocess
        port = os.environ.get("MASTER_PORT", "29541")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", port, str(Path(__file__).resolve())] + sys.argv[1:]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, cwd=str(ROOT))
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        for ln in r.stdout.splitlines():
            if not ln.startswith("{"):
                print(ln, file=sys.stderr)
        if lines:
            print(lines[-1], flush=True)
        raise SystemExit(r.returncode if r.returncode or lines else 1)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    # one process per GPU over RCCL ("nccl").  For a functional check of the multi-rank path on a single-GPU box:
    # WDG_DIST_BACKEND=gloo WDG_DEVICE=0 runs every rank on cuda:0 with host-staged reductions (not a benchmark).
    backend = os.environ.get("WDG_DIST_BACKEND", "nccl")
    dev_index = int(os.environ.get("WDG_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    # WDG_DIST_ALWAYS=1 with --gpus 1: a one-rank RCCL group, so the exchange code path (async all-reduce, deferred Adam,
    # SyncBN, metric reduce) runs on real collectives on a single-GPU box (functional check, not the headline)
    dist_on = world > 1 or os.environ.get("WDG_DIST_ALWAYS", "0") == "1"
    init_only = os.environ.get("WDG_DIST_INIT_ONLY", "0") == "1"      # A/B: the process group exists, the step does not use it
    if dist_on or init_only:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{dev_index}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from downscaling.data.data_generator import FlexibleNoiseGenerator
    from downscaling.engine import runtime
    from downscaling.gan import train
    from downscaling.gan.ganbase import GAN
    from downscaling.gan.models import make_discriminator, make_generator

    ops = runtime.get_ops()
    dev = ops.device
    for kv in args.tune:
        key, val = kv.split("=")
        if ops.lib.wdg_set_tuning(key.encode(), int(val)) != 0:
            raise SystemExit(f"--tune {kv}: rejected by wdg_set_tuning")
    B = args.batch
    generator = make_generator(S, CIN, NZ, CH, T)
    discriminator = make_discriminator(S, S, CIN, CH, T)
    gan = GAN(generator, discriminator, FlexibleNoiseGenerator((B, T, S, S, NZ), std=0.1, random_seed=1234, rank=rank),
              n_critic=3, distributed=dist_on, sync_bn=not args.no_sync_bn)
    gan.compile(generator_optimizer=train.generator_optimizer(), discriminator_optimizer=train.discriminator_optimizer(),
                discriminator_loss=train.discriminator_loss)
    low, high = synthetic_batch(B, 10 + rank, dev)
    if args.workload == "gen_fwd":
        return bench_generator_forward(args, generator, gan, low, world, rank, dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        gan.train_step((low, high))
    if os.environ.get("WDG_SYNC_DEBUG") == "1":       # (diagnostic: torch warns at every call that makes the host wait for the device)
        torch.cuda.set_sync_debug_mode("warn")
    timer = ConvTimer()
    timer.wrap(ops)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logs = gan.train_step((low, high))
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # host cost of enqueueing ONE step on an idle device (after a synchronize: no queue back-pressure in the figure; over many
    # back-to-back steps the host runs into the launch queue's depth and the same clock would read the device's time)
    concurrent_records = timer.records
    timer.records = []
    host_dt = None
    if not args.no_serial_pass:       # (profiling runs keep exactly warmup + steps train steps in their traces)
        t1 = time.perf_counter()
        gan.train_step((low, high))
        host_dt = time.perf_counter() - t1
        barrier()

    rccl_info = rccl_evidence(dist, gan, world, rank, dev) if dist_on else None
    # ---- per-kernel figures on ONE stream.  The default schedule runs the generator beside the discriminator, weight gradients
    # beside data gradients and the discriminator's two input branches beside each other on separate HIP streams: kernels of
    # different streams share the chip, so a per-launch duration inside the timed region (HIP events or a kernel trace alike)
    # measures the sharing, not the kernel.  The roofline of the dominant kernel is therefore taken from extra train steps with
    # every overlap switched off (same kernels, same launches, one stream), timed with the same per-launch HIP events; the
    # figures of the timed region are kept beside it ("concurrent").  Every rank runs the extra steps (collectives).
    timer.records = []
    saved = (gan.engine.overlap_generator, generator.net.wgrad_stream, discriminator.net.wgrad_stream,
             discriminator.net.overlap_branches)
    gan.engine.overlap_generator = False
    generator.net.wgrad_stream = discriminator.net.wgrad_stream = False
    discriminator.net.overlap_branches = False
    serial_steps = 2
    if args.no_serial_pass:
        serial_steps, dt_serial, timer.records = args.steps, dt, concurrent_records
    try:
        if not args.no_serial_pass:
            gan.train_step((low, high))
            timer.records = []
            barrier()
            t1 = time.perf_counter()
            for _ in range(serial_steps):
                gan.train_step((low, high))
            barrier()
            dt_serial = time.perf_counter() - t1
    finally:
        (gan.engine.overlap_generator, generator.net.wgrad_stream, discriminator.net.wgrad_stream,
         discriminator.net.overlap_branches) = saved
    serial_records = timer.records
    if rank == 0:
        # algorithmic FLOPs of one reference step per tile-timestep: 7*Gf + 28*Df (SURVEY §8 d; 240.5 GFLOP at S = 256, T = 1,
        # 41.7 at S = 96, T = 24), Gf / Df from the layer objects of the networks that ran
        gf, df = generator_flops(generator.net), discriminator_flops(discriminator.net)
        gf_exec, df_exec = generator_flops(generator.net, executed=True), discriminator_flops(discriminator.net, executed=True)
        step_flops = (7 * gf + 28 * df) * B * T
        timer.records = serial_records
        agg = timer.summary()
        timer.records = concurrent_records
        agg_conc = timer.summary()
        timer.records = serial_records
        if args.per_layer:
            for layer, kern, calls, ms, tf, tot in timer.per_layer(serial_steps):
                print(f"{tot:7.3f} ms/step  {calls:4.0f} x {ms:7.3f} ms  {tf:6.1f} TF/s  {kern:28s} {layer}", file=sys.stderr)
        dom = max(agg.items(), key=lambda kv: kv[1][1])
        conv_time = sum(a[1] for a in agg.values())
        conv_flops = sum(a[0] for a in agg.values())
        out = {
            "metric": "GAN train-step samples/s, 32x32->256x256 wind tiles" if headline else
                      f"GAN train-step tile-timesteps/s, {S // 8}x{S // 8}->{S}x{S} wind tiles, T={T}",
            "value": world * B * T * args.steps / dt,
            "unit": "samples/s" if headline else "tile-timesteps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "host_enqueue_ms_per_step": 1e3 * host_dt if host_dt is not None else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "GAN.train_step (G fwd x5, D fwd x12, 7 backward passes, Adam on both, n_critic=3, "
                                   f"metrics recompute) on G({S},3,20,2,T={T})+D({S},{S},3,2,T={T})" +
                                   (", configs[1]" if headline else " (not the headline configuration)"),
                       "per_gpu_batch": B, "global_batch": world * B, "image_size": S, "n_timesteps": T,
                       "parallelism": f"dp{world}" + ("" if args.no_sync_bn or not dist_on else "+syncbn")},
            "step_tflops_algorithmic": step_flops * 1e-12,
            "step_frac_of_mfma_f32_peak": step_flops / (dt / args.steps) / (PEAK_F32_MFMA_TFLOPS * 1e12),
            # the same on the multiply-adds the kernels EXECUTE: the upsample + 5x5 block runs in column form in all seven
            # generator passes, and at T = 1 (14.88 instead of 22.385 GFLOP per sample and pass at S = 256) the ConvLSTMs' dead
            # forget gate (c_0 = 0) is not computed in any direction
            "step_frac_executed": (7 * gf_exec + 28 * df_exec) * B * T / (dt / args.steps) / (PEAK_F32_MFMA_TFLOPS * 1e12),
            "roofline": {"bound": "mfma", "kernel": dom[0],
                         "achieved": dom[1][0] / dom[1][1] * 1e-12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": dom[1][0] / dom[1][1] * 1e-12 / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                         "launches": dom[1][2], "avg_launch_ms": 1e3 * dom[1][1] / dom[1][2],
                         "all_conv_kernels": {k: {"tflops": v[0] / v[1] * 1e-12, "ms_per_step": 1e3 * v[1] / serial_steps,
                                                   "launches_per_step": v[2] / serial_steps} for k, v in agg.items()},
                         "conv_share_of_step": conv_time / dt_serial, "all_conv_tflops": conv_flops / conv_time * 1e-12,
                         "measured": f"HIP events around every launch in {serial_steps} extra train steps on ONE stream (all overlaps off: "
                                     f"{1e3 * dt_serial / serial_steps:.2f} ms per step) after the timed region — under the default "
                                     "multi-stream schedule kernels of different streams share the chip and a per-launch duration "
                                     "measures the sharing, not the kernel; profiles/*_kernel_stats_serial.csv is the kernel trace of "
                                     "the same single-stream schedule (WDG_OVERLAP_GEN=0 WDG_WGRAD_STREAM=0 WDG_OVERLAP_BRANCHES=0)",
                         "single_stream_ms_per_step": 1e3 * dt_serial / serial_steps,
                         "concurrent": {"achieved": agg_conc[dom[0]][0] / agg_conc[dom[0]][1] * 1e-12 if dom[0] in agg_conc else None,
                                        "frac": agg_conc[dom[0]][0] / agg_conc[dom[0]][1] * 1e-12 / PEAK_F32_MFMA_TFLOPS if dom[0] in agg_conc else None,
                                        "avg_launch_ms": 1e3 * agg_conc[dom[0]][1] / agg_conc[dom[0]][2] if dom[0] in agg_conc else None,
                                        "note": "the same events inside the timed region (kernels of two or three streams in flight)"},
                         "measured_mfma_issue_ceiling_tflops": {"constant_operands": 155.7, "random_operands": 151.2,
                                                                "source": "tools/mfma_peak.hip, profiles/r01q_mfma_peak_probe.log"}},
            "losses": {k: float(v) for k, v in logs.items() if v is not None},
        }
        # HBM traffic of the dominant kernel comes from separate `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` passes
        # (counters cannot be read inside this process); the committed summary is attached only when it was collected
        # on THESE kernel sources (hash of csrc/) for the same kernel and the same number of launches per step —
        # otherwise `traffic` stays null rather than quoting a stale number.
        tj = ROOT / "profiles" / "pmc_traffic.json"
        if tj.exists():
            t = json.loads(tj.read_text())
            same = (t.get("kernel") == dom[0] and t.get("csrc_sha256") == csrc_hash()
                    and t.get("launches_per_step") == dom[1][2] / serial_steps)
            if same:
                out["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = t["source"]
            else:
                out["roofline"]["traffic_note"] = ("profiles/pmc_traffic.json was collected on other kernel sources / another "
                                                   "launch mix (kernel %s, %s launches per step) — not quoted" %
                                                   (t.get("kernel"), t.get("launches_per_step")))
        # the same kernel's figure from the rocprofv3 kernel trace of the one-stream schedule (tools/profile_round.sh ->
        # profiles/rocprof_dominant.json, same staleness guard): the tracer's clock reads a few percent lower than the
        # un-profiled HIP events — both are printed ("keep the dominant kernel's line honest")
        rj = ROOT / "profiles" / "rocprof_dominant.json"
        if rj.exists():
            r = json.loads(rj.read_text())
            if r.get("kernel") == dom[0] and r.get("csrc_sha256") == csrc_hash() and r.get("launches_per_step") == dom[1][2] / serial_steps:
                flops_per_launch = out["roofline"]["achieved"] * 1e12 * out["roofline"]["avg_launch_ms"] * 1e-3
                # the tracer's figure for the TIMED step of its run (its --stats mean also covers the warm-up step, whose launches
                # run at ramping clocks: profiles/rocprof_dominant.json keeps both, and the HIP-event figure of the same process)
                us = r.get("avg_launch_us_timed_step", r["avg_launch_us"])
                tf = flops_per_launch / (us * 1e-6) * 1e-12
                out["roofline"]["rocprof"] = {"avg_launch_us": us, "avg_launch_us_incl_warmup_step": r["avg_launch_us"],
                                              "hip_events_us_same_traced_process": r.get("hip_events_us"),
                                              "achieved": tf, "frac": tf / PEAK_F32_MFMA_TFLOPS, "source": r["source"]}
                if tf < out["roofline"]["achieved"]:
                    # quote the LOWER of the two clocks as the line's figure; the live HIP-event figure stays beside it
                    out["roofline"]["hip_events"] = {"achieved": out["roofline"]["achieved"], "frac": out["roofline"]["frac"]}
                    out["roofline"]["achieved"], out["roofline"]["frac"] = tf, tf / PEAK_F32_MFMA_TFLOPS
                    out["roofline"]["frac_source"] = "rocprofv3 kernel trace of the same one-stream schedule (lower than the live HIP events)"
        if headline and world == 1 and not args.no_generator_leg:
            out["generator_fwd_b64"] = generator_leg(generator, gan, dev)
        if headline and world == 1 and not args.no_config_legs:
            try:
                out["other_configs"] = other_config_legs(dev, ops)
            except Exception as exc:      # the extra legs must never cost the headline line
                out["other_configs"] = {"error": repr(exc)}
        if headline and world == 1 and not dist_on and not args.no_child_legs:
            out.update(child_legs(out["ms_per_step"]))
        if rccl_info is not None:
            out["rccl"] = rccl_info
        if getattr(ops, "_cstreams_probe", None):
            # how the generator / twin-discriminator streams were placed (HipOps.concurrent_streams: measured, not assumed)
            out["stream_placement"] = dict(ops._cstreams_probe, gpu_max_hw_queues=os.environ.get("GPU_MAX_HW_QUEUES", "default (4)"))
        if world == 1 and not args.no_cpu_baseline and headline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
