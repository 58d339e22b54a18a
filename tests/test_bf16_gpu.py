"""Inference precision (bf16 or IEEE-fp16 operands, fp32 accumulation) — BASELINE configs[3] / configs[4].  The reference is fp32-only, so
the tolerance is defined here: the bf16 kernels must equal an emulation that rounds the operands to bf16 and
accumulates exactly (rel 1e-4), and the bf16 generator forward must stay within 3e-2 (relative to the output
scale) of the fp32 oracle."""
import pytest
import torch

from oracle import torch_model as TM
from tests.helpers import randomize, rel_err

pytestmark = pytest.mark.gpu

CASES = [
    ("g0_8x8s2_cin23", 2, 32, 32, 23, 128, 8, 2, 3),
    ("g2_4x4s2", 2, 16, 16, 128, 128, 4, 2, 1),
    ("gates_128to512", 2, 16, 16, 128, 512, 3, 1, 1),
    ("c5_128to64", 2, 16, 16, 128, 64, 3, 1, 1),
    ("convT2x2_as_conv", 2, 16, 16, 32, 192, 2, 2, 0),
    # ... on a map the patch kernel tiles: its transposed direction (the generator's 2 x 2 stride-2 Conv2DTranspose) runs as ONE
    # GEMM with 4 * 32 columns and a scattering epilogue (WdgPatchH16::shufC)
    ("convT2x2_shuffle", 2, 48, 48, 32, 192, 2, 2, 0),
    ("convT2x2_shuffle_c20", 3, 32, 96, 20, 64, 2, 2, 0),
    ("odd_sizes", 3, 21, 19, 40, 72, 3, 1, 1),
    # conv_patch_h16.hip (input patch in LDS): 1 x 16 fragments (map width % 16) and 4 x 4 fragments (map width % 24)
    ("patchA_c0_96", 2, 96, 96, 23, 128, 8, 2, 3),
    ("patchA_c40_cout72", 2, 16, 32, 40, 72, 3, 1, 1),
    ("patchA_5x5_cout32", 1, 16, 16, 24, 32, 5, 1, 2),
    ("patchB_gates_24", 3, 24, 24, 128, 512, 3, 1, 1),
    ("patchB_c5_24", 2, 24, 24, 128, 64, 3, 1, 1),
    ("patchB_4x4s2_c16", 2, 48, 48, 16, 72, 4, 2, 1),
    ("patchC_c2_48", 2, 48, 48, 128, 128, 4, 2, 1),          # 4 x 24 tiles: the 8-row patch of 128 channels does not fit
    # 1 x 1: the transposed direction is the column GEMM of the column-form upsample layer (400 columns), through the patch kernel
    ("col_gemm_1x1", 2, 16, 32, 400, 160, 1, 1, 0),
    ("col_gemm_1x1_24", 2, 24, 24, 100, 64, 1, 1, 0),
]


@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_bf16_conv_kernels(case, fmt, hip_ops, ref_ops):
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    name, n, H, W, cin, cout, k, s, p = case
    gen = torch.Generator().manual_seed(3)
    dev = hip_ops.device
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    cin_p, cout_p = (cin + 3) // 4 * 4, (cout + 3) // 4 * 4
    x = torch.zeros(n, H, W, cin_p, dtype=torch.float64)
    x[..., :cin] = torch.randn(n, H, W, cin, generator=gen, dtype=torch.float64)
    dy = torch.zeros(n, Ho, Wo, cout_p, dtype=torch.float64)
    dy[..., :cout] = torch.randn(n, Ho, Wo, cout, generator=gen, dtype=torch.float64)
    w = torch.randn(k, k, cin, cout, generator=gen, dtype=torch.float64) * 0.05
    b = torch.randn(cout, generator=gen, dtype=torch.float64)
    aff = torch.cat([torch.rand(cout, generator=gen, dtype=torch.float64) + 0.5, torch.randn(cout, generator=gen, dtype=torch.float64)])
    g, rg = ConvGeom(k, k, s, p), RG(k, k, s, p)
    pk_g, pk_r = hip_ops.pack_weights(w.float().to(dev).contiguous()), ref_ops.pack_weights(w)
    if cin_p % 8 == 0:
        y_r, y_g = torch.zeros(n, Ho, Wo, cout_p, dtype=torch.float64), hip_ops.zeros(n, Ho, Wo, cout_p)
        ref_ops.conv_fwd_bf16(x.float().double(), pk_r, b, y_r, rg, act=True, affine=aff, fmt=fmt)
        hip_ops.conv_fwd_bf16(x.float().to(dev), pk_g, b.float().to(dev), y_g, g, act=True, affine=aff.float().to(dev), fmt=fmt)
        assert rel_err(y_g, y_r) < 1e-4, "fwd"
        ref_ops.conv_fwd_bf16(x.float().double(), pk_r, None, y_r, rg, accumulate=True, fmt=fmt)
        hip_ops.conv_fwd_bf16(x.float().to(dev), pk_g, None, y_g, g, accumulate=True, fmt=fmt)
        assert rel_err(y_g, y_r) < 1e-4, "fwd accumulate"
    if cout_p % 8 == 0:
        bi = torch.linspace(-1, 1, cin, dtype=torch.float64)
        dx_r, dx_g = torch.zeros(n, H, W, cin_p, dtype=torch.float64), hip_ops.zeros(n, H, W, cin_p)
        ref_ops.conv_dgrad_bf16(dy.float().double(), pk_r, dx_r, rg, bias=bi, act=True, fmt=fmt)
        hip_ops.conv_dgrad_bf16(dy.float().to(dev), pk_g, dx_g, g, bias=bi.float().to(dev), act=True, fmt=fmt)
        assert rel_err(dx_g, dx_r) < 1e-4, "transposed"


@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
def test_bf16_halo_kernels(fmt, hip_ops, ref_ops):
    """bf16 fused upsample + transposed 5x5 conv (two-stage LDS staging) and the thin 16 -> 2 conv."""
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    gen = torch.Generator().manual_seed(9)
    dev = hip_ops.device
    x = torch.randn(3, 60, 68, 160, generator=gen, dtype=torch.float64).float().double()
    w = torch.randn(5, 5, 16, 160, generator=gen, dtype=torch.float64) * 0.05
    b = torch.randn(16, generator=gen, dtype=torch.float64)
    aff = torch.cat([torch.rand(16, generator=gen, dtype=torch.float64) + 0.5, torch.randn(16, generator=gen, dtype=torch.float64)])
    y_r, y_g = torch.zeros(3, 120, 136, 16, dtype=torch.float64), hip_ops.zeros(3, 120, 136, 16)
    pk_g = hip_ops.pack_weights(w.float().to(dev).contiguous())
    ref_ops.upconv_colfwd = hip_ops.upconv_colfwd = False
    try:
        ref_ops.upconv_fwd_bf16(x, ref_ops.pack_weights(w), b, y_r, RG(5, 5, 1, 2), act=True, affine=aff, fmt=fmt)
        hip_ops.upconv_fwd_bf16(x.float().to(dev), pk_g, b.float().to(dev), y_g,
                                ConvGeom(5, 5, 1, 2), act=True, affine=aff.float().to(dev), fmt=fmt)
    finally:
        ref_ops.upconv_colfwd = hip_ops.upconv_colfwd = True
    # the interpolated value is rounded to bf16 after fp32 (HIP) vs fp64 (oracle) interpolation: a value that sits
    # on a rounding boundary may round the other way, so the bound is one bf16 ulp of a few of the 4000 products
    assert rel_err(y_g, y_r) < 2e-3
    # column form (default): the 16-bit operands are the low-res input and the weights, products and interpolation
    # exact / fp32 -> the same tight bound as the plain 16-bit convolutions
    y_r2, y_g2 = torch.zeros(3, 120, 136, 16, dtype=torch.float64), hip_ops.zeros(3, 120, 136, 16)
    ref_ops.upconv_fwd_bf16(x, ref_ops.pack_weights(w), b, y_r2, RG(5, 5, 1, 2), act=True, affine=aff, fmt=fmt)
    hip_ops.upconv_fwd_bf16(x.float().to(dev), pk_g, b.float().to(dev), y_g2, ConvGeom(5, 5, 1, 2), act=True,
                            affine=aff.float().to(dev), fmt=fmt)
    assert rel_err(y_g2, y_r2) < 1e-4
    assert rel_err(y_g2, y_g) < 1e-2    # both are 16-bit approximations of the same layer
    # the column form as ONE launch (csrc/upconv_fused_h16.hip, default for 160 -> 16) against its two-launch route: same rounding
    # points, fp32 sums in another order; on a map smaller than a tile row / column multiple and with pixel strides beyond C
    assert hip_ops.upconv_fused16 and hip_ops.lib.wdg_upconv_fused_h16_supported(160, 16)
    xr = torch.randn(2, 13, 21, 160, generator=gen, dtype=torch.float64).float()
    both = {}
    for fused in (True, False):
        hip_ops.upconv_fused16 = fused
        try:
            yb = hip_ops.zeros(2, 26, 42, 24)
            hip_ops.upconv_fwd_bf16(xr.to(dev), pk_g, b.float().to(dev), yb, ConvGeom(5, 5, 1, 2), act=True,
                                    affine=aff.float().to(dev), fmt=fmt)
        finally:
            hip_ops.upconv_fused16 = True
        assert float(yb[..., 16:].abs().max()) == 0.0
        both[fused] = yb
    y_rr = torch.zeros(2, 26, 42, 16, dtype=torch.float64)
    ref_ops.upconv_fwd_bf16(xr.double(), ref_ops.pack_weights(w), b, y_rr, RG(5, 5, 1, 2), act=True, affine=aff, fmt=fmt)
    assert rel_err(both[True][..., :16], y_rr) < 1e-4 and rel_err(both[False][..., :16], y_rr) < 1e-4
    assert rel_err(both[True], both[False]) < 1e-5
    # opt-in (WDG_Z16=1 / ops.z16): z itself stored in the 16-bit format.  The fp32 (HIP) and fp64 (oracle) sums can round to
    # different 16-bit neighbours: one 16-bit ulp of a few of the 100 terms of an output.  The 60 x 68 map above is outside the
    # patch kernel's tile shapes -> the switch must fall back to the fp32-z route bit-identically; 64 x 48 takes the 16-bit route
    saved = (getattr(ref_ops, "z16", False), hip_ops.z16)
    ref_ops.z16 = hip_ops.z16 = True
    try:
        y_g3 = hip_ops.zeros(3, 120, 136, 16)
        hip_ops.upconv_fwd_bf16(x.float().to(dev), pk_g, b.float().to(dev), y_g3, ConvGeom(5, 5, 1, 2), act=True,
                                affine=aff.float().to(dev), fmt=fmt)
        assert torch.equal(y_g3, y_g2)
        xs = torch.randn(2, 64, 48, 160, generator=gen, dtype=torch.float64).float().double()
        res = {}
        for z16, bound in ((False, 1e-4), (True, 2e-3)):
            ref_ops.z16 = hip_ops.z16 = z16
            y_r4, y_g4 = torch.zeros(2, 128, 96, 16, dtype=torch.float64), hip_ops.zeros(2, 128, 96, 16)
            ref_ops.upconv_fwd_bf16(xs, ref_ops.pack_weights(w), b, y_r4, RG(5, 5, 1, 2), act=True, affine=aff, fmt=fmt)
            hip_ops.upconv_fwd_bf16(xs.float().to(dev), pk_g, b.float().to(dev), y_g4, ConvGeom(5, 5, 1, 2), act=True,
                                    affine=aff.float().to(dev), fmt=fmt)
            assert rel_err(y_g4, y_r4) < bound, (z16, rel_err(y_g4, y_r4))
            res[z16] = y_g4
        assert 0 < rel_err(res[True], res[False]) < 1e-2
    finally:
        ref_ops.z16, hip_ops.z16 = saved
    x2 = torch.randn(2, 70, 50, 16, generator=gen, dtype=torch.float64).float().double()
    w2 = torch.randn(3, 3, 16, 2, generator=gen, dtype=torch.float64) * 0.2
    b2 = torch.randn(2, generator=gen, dtype=torch.float64)
    o_r, o_g = torch.zeros(2, 70, 50, 4, dtype=torch.float64), hip_ops.zeros(2, 70, 50, 4)
    ref_ops.conv_halo_fwd_bf16(x2, ref_ops.pack_weights(w2), b2, o_r, RG(3, 3, 1, 1), fmt=fmt)
    hip_ops.conv_halo_fwd_bf16(x2.float().to(dev), hip_ops.pack_weights(w2.float().to(dev).contiguous()), b2.float().to(dev),
                               o_g, ConvGeom(3, 3, 1, 1), fmt=fmt)
    assert rel_err(o_g, o_r) < 1e-4
    assert float(o_g[..., 2:].abs().max()) == 0.0


@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
@pytest.mark.parametrize("S,T,F", [(32, 2, 128), (48, 1, 64)])
def test_bf16_generator_forward(hip_ops, S, T, F, fmt):
    from downscaling.engine.networks import GeneratorNet
    B, cin, nz, ch = 2, 3, 20, 2
    dev = hip_ops.device
    net = GeneratorNet(hip_ops, S, cin, nz, ch, T, feature_channels=F, seed=3)
    w = randomize(net, 11)
    g = torch.Generator().manual_seed(0)
    low = torch.randn(B, T, S, S, cin, generator=g, dtype=torch.float64)
    noise = torch.randn(B, T, S, S, nz, generator=g, dtype=torch.float64) * 0.1
    net.set_image(low.float().to(dev))
    net.set_noise(noise.float().to(dev))
    out16 = torch.zeros(B, T, S, S, ch, device=dev)
    net.from_time_major(net.forward(B, False, precision=fmt), out16)
    out32 = torch.zeros(B, T, S, S, ch, device=dev)
    net.from_time_major(net.forward(B, False, precision="fp32"), out32)
    ref = TM.generator_forward(w, low, noise, False)
    assert rel_err(out32, ref) < 1e-4
    err = rel_err(out16, ref)
    assert 1e-6 < err < (3e-2 if fmt == "bf16" else 4e-3), err   # a different precision, within the stated tolerance
    # the activation between the last two layers handed over in the operand format (WDG_ACT16, default) or in fp32: the reader
    # rounds to the operand format either way — the outputs are the same bits
    if F == 128 and hip_ops.act16_output_conv_ok(net.c9.pk, net.c11.pk, net.c11.g):
        hip_ops.act16 = False
        try:
            out16b = torch.zeros(B, T, S, S, ch, device=dev)
            net.from_time_major(net.forward(B, False, precision=fmt), out16b)
        finally:
            hip_ops.act16 = True
        assert torch.equal(out16, out16b)
    with pytest.raises(ValueError):
        net.forward(B, True, precision=fmt)


@pytest.mark.parametrize("switch", ["upconv_colfwd", "upconv_fused16"])
def test_act16_buffers_follow_the_upsample_route(hip_ops, switch):
    """with the documented switches WDG_UPCONV_COLFWD=0 / WDG_UPCONV_FUSED16=0 the fused upsample kernel
    — the only route of upconv_fwd_bf16 that takes activations in the 16-bit operand format — is not taken, so the generator
    must not allocate 16-bit hand-over buffers (a 16-bit z9 in front of an fp32-writing route was a device out-of-bounds
    write), the forward must stay inside the stated tolerance, and a direct call with a 16-bit y must be refused."""
    from downscaling.engine.networks import GeneratorNet
    B, T, S, cin, nz, ch, F, fmt = 2, 1, 32, 3, 20, 2, 128, "bf16"
    dev = hip_ops.device
    saved = getattr(hip_ops, switch)
    setattr(hip_ops, switch, False)
    try:
        net = GeneratorNet(hip_ops, S, cin, nz, ch, T, feature_channels=F, seed=3)
        w = randomize(net, 11)
        g = torch.Generator().manual_seed(0)
        low = torch.randn(B, T, S, S, cin, generator=g, dtype=torch.float64)
        noise = torch.randn(B, T, S, S, nz, generator=g, dtype=torch.float64) * 0.1
        net.set_image(low.float().to(dev))
        net.set_noise(noise.float().to(dev))
        assert not hip_ops.act16_output_conv_ok(net.c9.pk, net.c11.pk, net.c11.g)
        out16 = torch.zeros(B, T, S, S, ch, device=dev)
        net.from_time_major(net.forward(B, False, precision=fmt), out16)
        b = net.buffers(B)
        assert "z9_" + fmt not in b and b.get("cat2_" + fmt) is None
        ref = TM.generator_forward(w, low, noise, False)
        assert 1e-6 < rel_err(out16, ref) < 3e-2
        y16 = torch.zeros(*b["z9"].shape, dtype=torch.bfloat16, device=dev)
        with pytest.raises(ValueError):
            hip_ops.upconv_fwd_bf16(b["cat2"], net.c9.pk, net.c9.b.value, y16, net.c9.g, act=True, fmt=fmt)
    finally:
        setattr(hip_ops, switch, saved)


@pytest.mark.parametrize("fmt", ["bf16", "fp16"])
@pytest.mark.parametrize("S,T,F", [(96, 3, 128), (64, 2, 64)])
def test_convlstm16_fused_step_matches_unfused(hip_ops, S, T, F, fmt):
    """16-bit ConvLSTM: one launch per timestep (recurrent convolution with the cell update in its epilogue, gate columns
    interleaved — wdg_convlstm_step_h16) against the three-launch form (accumulating convolution + wdg_lstm_fwd): the same
    rounding points and the same fp32 cell arithmetic, so the hidden states agree to fp32 accumulation-order noise."""
    from downscaling.engine.networks import GeneratorNet
    B, cin, nz, ch = 2, 3, 20, 2
    dev = hip_ops.device
    net = GeneratorNet(hip_ops, S, cin, nz, ch, T, feature_channels=F, seed=3)
    randomize(net, 11)
    g = torch.Generator().manual_seed(0)
    net.set_image(torch.randn(B, T, S, S, cin, generator=g).to(dev))
    net.set_noise((torch.randn(B, T, S, S, nz, generator=g) * 0.1).to(dev))
    outs = []
    b = net.buffers(B)
    for fused, act16 in ((1, True), (1, False), (0, True)):
        assert hip_ops.lib.wdg_set_tuning(b"lstm16_fused", fused) == 0
        hip_ops.act16 = act16
        try:
            b["h"].fill_(7.0)                 # (a route that does not write the fp32 state must not be credited with an older one)
            if b.get("h_" + fmt) is not None:
                b["h_" + fmt].fill_(7.0)
            out = net.forward(B, False, precision=fmt).clone()
            outs.append((out, b["h"].clone(), b["h_" + fmt].clone() if b.get("h_" + fmt) is not None else None))
        finally:
            hip_ops.lib.wdg_set_tuning(b"lstm16_fused", 1)
            hip_ops.act16 = True
    x = b["cat4"][..., F // 2:]
    assert hip_ops.convlstm16_supported(x, net.lstm.gates, net.lstm.pkx, net.lstm.g, F), "the fused path must be the one tested"
    # (1) the hidden state (and the quarter-resolution activations) kept in the operand format — the default — against the fp32
    # hand-over: every reader rounds to the operand format while staging, so the state IS the rounded fp32 state and the generator's
    # output is the same bits
    if (S, F) == (96, 128):
        assert outs[0][2] is not None, "the shipped shape keeps the ConvLSTM state in the operand format"
    if outs[0][2] is not None:
        assert float(outs[0][1].min()) == 7.0, "... and only there"
        assert torch.equal(outs[0][2], outs[1][1].to(outs[0][2].dtype)), "16-bit state == rounded fp32 state"
    assert torch.equal(outs[0][0], outs[1][0]), "same output bits with either hand-over"
    # (2) fused step against the three-launch form (which keeps fp32 buffers: the switch moves the layer off the 16-bit-state route)
    assert rel_err(outs[1][1], outs[2][1]) < 2e-5, "hidden states"
    # behind the ConvLSTM every 16-bit layer re-rounds its input: a 1e-7 difference in h flips an occasional operand by one 16-bit
    # ulp (bf16 4e-3, fp16 5e-4 relative), so the outputs agree to a few operand ulps at isolated pixels, not to fp32 noise
    assert rel_err(outs[1][0], outs[2][0]) < (1e-2 if fmt == "bf16" else 2e-3), "generator output"
