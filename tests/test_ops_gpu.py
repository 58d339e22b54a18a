"""Operator parity on the GPU: every entry point of libwdgan.so (through the C ABI / HipOps) against
the float64 CPU oracle (oracle/torch_backend.py) on identical seeded inputs.

Tolerance: the HIP path computes in exact fp32 (v_mfma_f32_16x16x4_f32 is an fp32 fma chain), so the
only difference to the fp64 oracle is fp32 rounding: max|err| <= 2e-5 * max|ref| (K up to 12,544),
far inside the 1e-4 relative target of BASELINE.json.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-5


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def both(hip_ops, t64):
    """float64 CPU tensor -> (cpu64, gpu32) pair."""
    return t64, t64.float().to(hip_ops.device)


CONV_CASES = [
    # name, n, H, W, cin, cout, k, s, p
    ("g5_3x3", 2, 16, 16, 128, 64, 3, 1, 1),
    ("g0_8x8s2_cin23", 2, 32, 32, 23, 128, 8, 2, 3),
    ("g2_4x4s2", 2, 16, 16, 128, 128, 4, 2, 1),
    ("d_7x7s3", 3, 20, 20, 32, 64, 7, 3, 1),
    ("d_7x7s3_odd", 2, 27, 27, 64, 128, 7, 3, 1),
    ("d_3x3s2_valid", 4, 3, 3, 256, 512, 3, 2, 0),
    ("lstm_2to8", 2, 24, 24, 2, 8, 3, 1, 1),
    ("lstm_5to64", 2, 24, 24, 5, 64, 3, 1, 1),
    ("out_16to2", 2, 24, 24, 16, 2, 3, 1, 1),
    ("convT2x2_as_conv", 2, 16, 16, 32, 192, 2, 2, 0),
    ("convT5x5_as_conv", 2, 16, 16, 16, 160, 5, 1, 2),
    ("multi_tile", 4, 64, 64, 128, 128, 3, 1, 1),
    ("splitk_7x7_small_m", 8, 8, 8, 256, 512, 7, 3, 1),
    ("gates_128to512", 2, 16, 16, 128, 512, 3, 1, 1),
    # >= 65536 output pixels, stride 1, few channels: these take the halo-tile kernel (conv_halo.hip)
    ("halo_lstm_2to8", 5, 120, 136, 2, 8, 3, 1, 1),
    ("halo_lstm_5to64", 4, 128, 128, 5, 64, 3, 1, 1),
    ("halo_16to16", 5, 120, 136, 16, 16, 3, 1, 1),
    ("halo_out_16to2", 4, 128, 128, 16, 2, 3, 1, 1),
    ("halo_convT5x5_as_conv", 4, 128, 130, 16, 160, 5, 1, 2),
    ("halo_3x3_128to64", 4, 128, 128, 128, 64, 3, 1, 1),
    ("halo_5x5_8to16", 4, 128, 128, 8, 16, 5, 1, 2),
    ("halo_3x3_12to24", 4, 128, 132, 12, 24, 3, 1, 1),
    ("n160_tile32", 2, 24, 24, 16, 160, 5, 1, 2),
    # 65,536 .. 131,071 output pixels: fewer than two 8-row halo tiles per CU -> 4-row tiles (the per-timestep recurrent
    # convolutions of the discriminator's ConvLSTMs at the shipped shape: 8 tiles of 96 x 96)
    ("halo_th4_16to64", 8, 96, 96, 16, 64, 3, 1, 1),
    ("halo_th4_2to8", 8, 96, 96, 2, 8, 3, 1, 1),
    ("halo_th4_ragged", 9, 90, 100, 16, 64, 3, 1, 1),
    # thin 3x3 weight-gradient kernel: ragged tile edges, every built row-tile count, 32 output channels
    ("thin_3to32", 3, 150, 150, 3, 32, 3, 1, 1),
    ("thin_4to8", 3, 150, 151, 4, 8, 3, 1, 1),
    ("thin_8to16", 3, 149, 150, 8, 16, 3, 1, 1),
    ("thin_1to4", 5, 120, 130, 1, 4, 3, 1, 1),
]


def _mk(case, seed=0):
    name, n, H, W, cin, cout, k, s, p = case
    g = torch.Generator().manual_seed(seed)
    Ho = (H + 2 * p - k) // s + 1
    Wo = (W + 2 * p - k) // s + 1
    cin_p, cout_p = (cin + 3) // 4 * 4, (cout + 3) // 4 * 4
    x = torch.zeros(n, H, W, cin_p, dtype=torch.float64)
    x[..., :cin] = torch.randn(n, H, W, cin, generator=g, dtype=torch.float64)
    dy = torch.zeros(n, Ho, Wo, cout_p, dtype=torch.float64)
    dy[..., :cout] = torch.randn(n, Ho, Wo, cout, generator=g, dtype=torch.float64)
    w = torch.randn(k, k, cin, cout, generator=g, dtype=torch.float64) * 0.05
    b = torch.randn(cout, generator=g, dtype=torch.float64)
    return dict(n=n, H=H, W=W, Ho=Ho, Wo=Wo, cin=cin, cout=cout, cin_p=cin_p, cout_p=cout_p, k=k, s=s, p=p,
                x=x, dy=dy, w=w, b=b)


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(case, hip_ops, ref_ops):
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    d = _mk(case)
    g, rg = ConvGeom(d["k"], d["k"], d["s"], d["p"]), RG(d["k"], d["k"], d["s"], d["p"])
    dev = hip_ops.device
    w_g = d["w"].float().to(dev).contiguous()
    pk_g, pk_r = hip_ops.pack_weights(w_g), ref_ops.pack_weights(d["w"])
    x_g, dy_g, b_g = d["x"].float().to(dev), d["dy"].float().to(dev), d["b"].float().to(dev)

    # forward: bias + LeakyReLU, then accumulate on top of a non-zero output
    y_r = torch.zeros(d["n"], d["Ho"], d["Wo"], d["cout_p"], dtype=torch.float64)
    y_g = torch.zeros_like(y_r, dtype=torch.float32, device=dev)
    ref_ops.conv_fwd(d["x"], pk_r, d["b"], y_r, rg, act=True)
    hip_ops.conv_fwd(x_g, pk_g, b_g, y_g, g, act=True)
    assert rel_err(y_g, y_r) < TOL, "fwd"
    if d["cout_p"] != d["cout"]:
        assert float(y_g[..., d["cout"]:].abs().max()) == 0.0, "pad channels must stay zero"
    ref_ops.conv_fwd(d["x"], pk_r, None, y_r, rg, act=False, accumulate=True)
    hip_ops.conv_fwd(x_g, pk_g, None, y_g, g, act=False, accumulate=True)
    assert rel_err(y_g, y_r) < TOL, "fwd accumulate"

    # data gradient (also the Conv2DTranspose forward when bias/act are given)
    dx_r = torch.zeros(d["n"], d["H"], d["W"], d["cin_p"], dtype=torch.float64)
    dx_g = torch.zeros_like(dx_r, dtype=torch.float32, device=dev)
    ref_ops.conv_dgrad(d["dy"], pk_r, dx_r, rg)
    hip_ops.conv_dgrad(dy_g, pk_g, dx_g, g)
    assert rel_err(dx_g, dx_r) < TOL, "dgrad"
    bi_r = torch.linspace(-1, 1, d["cin"], dtype=torch.float64)
    ref_ops.conv_dgrad(d["dy"], pk_r, dx_r, rg, bias=bi_r, act=True, accumulate=True)
    hip_ops.conv_dgrad(dy_g, pk_g, dx_g, g, bias=bi_r.float().to(dev), act=True, accumulate=True)
    assert rel_err(dx_g, dx_r) < TOL, "dgrad bias/act/accumulate"

    # weight gradient, overwrite then accumulate
    dw_r = torch.zeros_like(d["w"])
    dw_g = torch.full_like(w_g, 7.0)
    ref_ops.conv_wgrad(d["x"], d["dy"], pk_r, dw_r, rg, accumulate=False)
    hip_ops.conv_wgrad(x_g, dy_g, pk_g, dw_g, g, accumulate=False)
    assert rel_err(dw_g, dw_r) < TOL, "wgrad"
    ref_ops.conv_wgrad(d["x"], d["dy"], pk_r, dw_r, rg, accumulate=True)
    hip_ops.conv_wgrad(x_g, dy_g, pk_g, dw_g, g, accumulate=True)
    assert rel_err(dw_g, dw_r) < TOL, "wgrad accumulate"
    # weight + bias gradient in one call (thin 3x3 layers: ones-row by-product; others: column-sum pass)
    db_r = torch.linspace(-2, 2, d["cout"], dtype=torch.float64)
    db_g = db_r.float().to(dev)
    ref_ops.conv_wgrad(d["x"], d["dy"], pk_r, dw_r, rg, accumulate=True, dbias=db_r)
    hip_ops.conv_wgrad(x_g, dy_g, pk_g, dw_g, g, accumulate=True, dbias=db_g)
    assert rel_err(dw_g, dw_r) < TOL, "wgrad+bias: dw"
    assert rel_err(db_g, db_r) < TOL, "wgrad+bias: dbias"


@pytest.mark.parametrize("n,H,W,cin,F_,k,s,p", [(3, 20, 24, 128, 128, 3, 1, 1), (2, 17, 19, 32, 16, 3, 1, 1), (2, 30, 28, 64, 32, 4, 2, 1)])
def test_conv_gradients_over_channel_ranges(n, H, W, cin, F_, k, s, p, hip_ops, ref_ops):
    """wdg_conv_plan_create_sliced (HipOps.conv_dgrad_slice / conv_wgrad_slice): the data and weight gradient of the LIVE gate
    columns of a ConvLSTM2D at n_timesteps = 1 (models.py:45 — gates i, c~, o = channel ranges [0, F) and [2F, 4F) of the 4F-channel
    gate tensor; the forget gate's gradient is exactly zero) against the full-width gradients of the oracle on a dgates tensor
    whose forget-gate channels are zero.  The forget-gate columns of dW must be left untouched."""
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    d = _mk(("ranges", n, H, W, cin, 4 * F_, k, s, p), seed=3)
    d["dy"][..., F_:2 * F_] = 0
    g, rg = ConvGeom(k, k, s, p), RG(k, k, s, p)
    dev = hip_ops.device
    w_g = d["w"].float().to(dev).contiguous()
    pk_g, pk_r = hip_ops.pack_weights(w_g), ref_ops.pack_weights(d["w"])
    x_g, dy_g = d["x"].float().to(dev), d["dy"].float().to(dev)
    dx_r = torch.zeros(n, H, W, d["cin_p"], dtype=torch.float64)
    ref_ops.conv_dgrad(d["dy"], pk_r, dx_r, rg)
    dx_g = torch.full((n, H, W, d["cin_p"]), 3.0, dtype=torch.float32, device=dev)
    hip_ops.conv_dgrad_slice(dy_g[..., :F_], pk_g, 0, F_, dx_g, g, accumulate=False)
    hip_ops.conv_dgrad_slice(dy_g[..., 2 * F_:], pk_g, 2 * F_, 4 * F_, dx_g, g, accumulate=True)
    assert rel_err(dx_g, dx_r) < TOL, "dgrad over the live ranges"
    dw_r = torch.zeros_like(d["w"])
    ref_ops.conv_wgrad(d["x"], d["dy"], pk_r, dw_r, rg, accumulate=False)
    dw_g = torch.full_like(w_g, 7.0)
    hip_ops.conv_wgrad_slice(x_g, dy_g[..., :F_], pk_g, 0, F_, dw_g, g, accumulate=False)
    hip_ops.conv_wgrad_slice(x_g, dy_g[..., 2 * F_:], pk_g, 2 * F_, 4 * F_, dw_g, g, accumulate=False)
    assert float((dw_g[..., F_:2 * F_] - 7.0).abs().max()) == 0.0, "forget-gate columns of dW untouched"
    live = torch.cat([dw_g[..., :F_], dw_g[..., 2 * F_:]], -1)
    live_r = torch.cat([dw_r[..., :F_], dw_r[..., 2 * F_:]], -1)
    assert rel_err(live, live_r) < TOL, "wgrad over the live ranges"
    hip_ops.conv_wgrad_slice(x_g, dy_g[..., :F_], pk_g, 0, F_, dw_g, g, accumulate=True)
    assert rel_err(dw_g[..., :F_], 2 * dw_r[..., :F_]) < TOL, "wgrad accumulate"


def test_conv_on_concat_and_time_views(hip_ops, ref_ops):
    """Zero-copy channel concat (ld > C, channel offset) and time-sliced image stride."""
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    g, rg = ConvGeom(3, 3, 1, 1), RG(3, 3, 1, 1)
    gen = torch.Generator().manual_seed(5)
    dev = hip_ops.device
    buf = torch.randn(2, 3, 12, 12, 192, generator=gen, dtype=torch.float64)  # [B,T,H,W,C]
    w = torch.randn(3, 3, 128, 32, generator=gen, dtype=torch.float64) * 0.05
    buf_g = buf.float().to(dev)
    x_r, x_g = buf[:, 1, :, :, 64:192], buf_g[:, 1, :, :, 64:192]
    out = torch.zeros(2, 12, 12, 64, dtype=torch.float64)
    out_g = out.float().to(dev)
    pk_r, pk_g = ref_ops.pack_weights(w), hip_ops.pack_weights(w.float().to(dev).contiguous())
    ref_ops.conv_fwd(x_r, pk_r, None, out[..., 32:64], rg)
    hip_ops.conv_fwd(x_g, pk_g, None, out_g[..., 32:64], g)
    assert rel_err(out_g, out) < TOL
    assert float(out_g[..., :32].abs().max()) == 0.0
    # dgrad accumulating into the strided concat view
    ref_ops.conv_dgrad(out[..., 32:64], pk_r, x_r, rg, accumulate=True)
    hip_ops.conv_dgrad(out_g[..., 32:64], pk_g, x_g, g, accumulate=True)
    assert rel_err(buf_g, buf) < TOL


@pytest.mark.parametrize("n,H,W,C,N,ld", [(3, 60, 68, 160, 16, 160), (2, 19, 35, 40, 4, 48), (2, 3, 3, 20, 8, 20),
                                          (1, 9, 50, 160, 16, 192), (2, 4, 17, 8, 2, 8)])
@pytest.mark.parametrize("composite", [True, False, "column"])
def test_fused_upsample_conv_transpose(n, H, W, C, N, ld, composite, hip_ops, ref_ops):
    """UpSampling2D(bilinear) + Conv2DTranspose(5x5,'same') + bias + LeakyReLU fused (models.py:62-64), input taken
    from a channel-concat view, ragged tiles, tiny maps (every pixel on the border ring), few outputs / channels.
    composite=True: the four-phase 4x4 composite-kernel path (upconv4.hip); False: the 25-tap halo kernel."""
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    g, rg = ConvGeom(5, 5, 1, 2), RG(5, 5, 1, 2)
    gen = torch.Generator().manual_seed(9)
    dev = hip_ops.device
    x = torch.randn(n, H, W, C, generator=gen, dtype=torch.float64)
    w = torch.randn(5, 5, N, C, generator=gen, dtype=torch.float64) * 0.05
    b = torch.randn(N, generator=gen, dtype=torch.float64)
    Np = (N + 3) // 4 * 4
    y_r = torch.zeros(n, 2 * H, 2 * W, Np, dtype=torch.float64)
    y_g = hip_ops.zeros(n, 2 * H, 2 * W, Np)
    ref_ops.upconv_fwd(x, ref_ops.pack_weights(w), b, y_r, rg, act=True)
    xg = hip_ops.zeros(n, H, W, ld)
    xg[..., :C] = x.float().to(dev)
    if ld > C:
        xg[..., C:] = 7.0   # neighbouring channels of the concat buffer must not leak in
    hip_ops.upconv4 = composite is True
    hip_ops.upconv_colfwd = composite == "column"     # 1x1 GEMM + bilinear gather (N = 2 falls back)
    try:
        hip_ops.upconv_fwd(xg[..., :C], hip_ops.pack_weights(w.float().to(dev).contiguous()), b.float().to(dev), y_g, g, act=True)
    finally:
        hip_ops.upconv4, hip_ops.upconv_colfwd = True, True
    assert rel_err(y_g, y_r) < TOL
    if Np != N:
        assert float(y_g[..., N:].abs().max()) == 0.0


@pytest.mark.parametrize("n,H,C,ld,k,s,p,t", [(3, 9, 128, 128, 6, 11, 4, 2), (2, 13, 8, 12, 7, 8, 5, 3), (2, 6, 16, 16, 6, 6, 0, 1),
                                              (1, 12, 4, 8, 6, 7, 4, 3)])
def test_window_patches_gather_and_adjoint(n, H, C, ld, k, s, p, t, hip_ops, ref_ops):
    """wdg_patch_gather / wdg_patch_scatter (shortcut_convolution as a 1x1 conv on disjoint windows, tf_utils.py:15-32):
    against the oracle's slicing loops, the scatter accumulating into a strided channel view; and the adjoint identity
    <gather(x), g> = <x, scatter(g)>."""
    gen = torch.Generator().manual_seed(3)
    dev = hip_ops.device
    x = torch.randn(n, H, H, C, generator=gen, dtype=torch.float64)
    g = torch.randn(n, t, t, k * k * C, generator=gen, dtype=torch.float64)
    base = torch.randn(n, H, H, C, generator=gen, dtype=torch.float64)
    out_r, dx_r = torch.zeros(n, t, t, k * k * C, dtype=torch.float64), base.clone()
    ref_ops.patch_gather(x, out_r, k, s, p)
    ref_ops.patch_scatter(g, dx_r, k, s, p, accumulate=True)
    xg = hip_ops.zeros(n, H, H, ld)
    xg[..., :C] = x.float().to(dev)
    out_g = hip_ops.empty(n, t, t, k * k * C)
    hip_ops.patch_gather(xg[..., :C], out_g, k, s, p)
    dxg = hip_ops.zeros(n, H, H, ld)
    dxg.fill_(5.0)
    dxg[..., :C] = base.float().to(dev)
    hip_ops.patch_scatter(g.float().to(dev).contiguous(), dxg[..., :C], k, s, p, accumulate=True)
    assert rel_err(out_g, out_r) < TOL and rel_err(dxg[..., :C], dx_r) < TOL
    if ld > C:
        assert float((dxg[..., C:] - 5.0).abs().max()) == 0.0
    lhs = float((out_r * g).sum())
    rhs = float((x * (dx_r - base)).sum())
    assert abs(lhs - rhs) < 1e-9 * max(1.0, abs(lhs))


@pytest.mark.parametrize("n,H,W,C,N,ld", [(3, 60, 68, 160, 16, 160), (2, 19, 35, 40, 4, 48), (2, 3, 3, 20, 8, 20),
                                          (1, 9, 50, 160, 16, 192), (2, 4, 17, 8, 2, 8), (1, 2, 2, 12, 16, 12)])
@pytest.mark.parametrize("column", [True, False])
def test_upsample_conv_transpose_backward(n, H, W, C, N, ld, column, hip_ops, ref_ops):
    """Backward of UpSampling2D(bilinear) + Conv2DTranspose(5x5,'same') (models.py:60-64): weight gradient
    (accumulated) and low-res input gradient written into a channel-concat view.  column=True: column form on the
    replicate-extended low-res grid (upconv_col.hip; N = 2 falls back), False: via the materialised upsampled tensor.
    Ragged tiles and maps so small that every pixel touches the border masks."""
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    g, rg = ConvGeom(5, 5, 1, 2), RG(5, 5, 1, 2)
    gen = torch.Generator().manual_seed(21)
    dev = hip_ops.device
    Np = (N + 3) // 4 * 4
    x = torch.randn(n, H, W, C, generator=gen, dtype=torch.float64)
    w = torch.randn(5, 5, N, C, generator=gen, dtype=torch.float64) * 0.05
    dy = torch.zeros(n, 2 * H, 2 * W, Np, dtype=torch.float64)
    dy[..., :N] = torch.randn(n, 2 * H, 2 * W, N, generator=gen, dtype=torch.float64)
    dw0 = torch.randn(5, 5, N, C, generator=gen, dtype=torch.float64)
    dw_r, dx_r = dw0.clone(), torch.zeros(n, H, W, C, dtype=torch.float64)
    ref_ops.upconv_bwd(x, dy, ref_ops.pack_weights(w), dw_r, dx_r, rg)
    xg = hip_ops.zeros(n, H, W, ld)
    xg[..., :C] = x.float().to(dev)
    dxg = hip_ops.zeros(n, H, W, ld)
    dxg.fill_(3.0)
    dw_g = dw0.float().to(dev).contiguous()
    hip_ops.upconv_col = column
    try:
        hip_ops.upconv_bwd(xg[..., :C], dy.float().to(dev), hip_ops.pack_weights(w.float().to(dev).contiguous()), dw_g,
                           dxg[..., :C], g)
    finally:
        hip_ops.upconv_col = True
    assert rel_err(dxg[..., :C], dx_r) < TOL
    assert rel_err(dw_g, dw_r) < TOL
    if ld > C:
        assert float((dxg[..., C:] - 3.0).abs().max()) == 0.0   # neighbouring channels of the concat buffer untouched


@pytest.mark.parametrize("cin,F_,n,H,W", [(2, 2, 3, 37, 45), (5, 16, 2, 33, 70), (5, 16, 1, 4, 32)])
def test_fused_single_step_convlstm(cin, F_, n, H, W, hip_ops, ref_ops):
    """convlstm1.hip: x -> h and (x, dh) -> (dgates, dx) with gate recomputation, against
    conv_fwd + lstm_fwd / lstm_bwd + conv_dgrad of the oracle backend; strided concat-style views."""
    gen = torch.Generator().manual_seed(13)
    dev = hip_ops.device
    cp, Fp = (cin + 3) // 4 * 4, (F_ + 3) // 4 * 4
    xb = torch.zeros(n, H, W, cp + 4, dtype=torch.float64)
    xb[..., :cin] = torch.randn(n, H, W, cin, generator=gen, dtype=torch.float64)
    wx = torch.randn(3, 3, cin, 4 * F_, generator=gen, dtype=torch.float64) * 0.4
    bias = torch.randn(4 * F_, generator=gen, dtype=torch.float64) * 0.3
    dh = torch.zeros(n, H, W, Fp, dtype=torch.float64)
    dh[..., :F_] = torch.randn(n, H, W, F_, generator=gen, dtype=torch.float64)
    res = {}
    for name, ops, cv in (("ref", ref_ops, lambda t: t.clone()), ("hip", hip_ops, lambda t: t.float().to(dev))):
        xx, ww, bb, dd = cv(xb), cv(wx).contiguous(), cv(bias), cv(dh)
        xv = xx[..., :cp]
        h = ops.zeros(n, H, W, Fp)
        ops.convlstm1_fwd(xv, ww, bb, h, cin, F_)
        dg = ops.empty(n, H, W, 4 * F_)
        dx = cv(torch.ones(n, H, W, cp, dtype=torch.float64))
        ops.convlstm1_bwd(xv, ww, bb, dd, dg, dx, cin, F_, accumulate_dx=True)
        dx2 = ops.zeros(n, H, W, cp)
        ops.convlstm1_bwd(xv, ww, bb, dd, None, dx2, cin, F_, accumulate_dx=False)
        # kernel + bias gradient fused into the same pass (accumulating onto existing values), with and without dx
        dw, db = cv(torch.full((3, 3, cin, 4 * F_), 0.5, dtype=torch.float64)).contiguous(), cv(torch.full((4 * F_,), -0.25, dtype=torch.float64))
        dx3 = ops.zeros(n, H, W, cp)
        ops.convlstm1_bwd(xv, ww, bb, dd, None, dx3, cin, F_, accumulate_dx=False, dw=dw, dbias=db)
        dw2, db2 = ops.zeros(3, 3, cin, 4 * F_), ops.zeros(4 * F_)
        ops.convlstm1_bwd(xv, ww, bb, dd, None, None, cin, F_, dw=dw2, dbias=db2)
        res[name] = dict(h=h, dg=dg, dx=dx, dx2=dx2, dx3=dx3, dw=dw, db=db, dw2=dw2, db2=db2)
        if cin == 5:
            # the last two input channels from a SECOND tensor (the high-res part of concat(low, high) read in place,
            # wdg_convlstm1_fwd_x2 / _bwd_x2): x carries garbage there
            n2 = 2
            x2 = ops.zeros(n, H, W, 4)
            x2[..., :n2] = xv[..., cin - n2:cin]
            xg = xv.clone()
            xg[..., cin - n2:cin] = 77.0
            assert ops.convlstm1_x2_supported(cin, F_, n2)
            h2 = ops.zeros(n, H, W, Fp)
            ops.convlstm1_fwd(xg, ww, bb, h2, cin, F_, x2=(x2, n2))
            dx4, dw4, db4 = ops.zeros(n, H, W, cp), ops.zeros(3, 3, cin, 4 * F_), ops.zeros(4 * F_)
            ops.convlstm1_bwd(xg, ww, bb, dd, None, dx4, cin, F_, dw=dw4, dbias=db4, x2=(x2, n2))
            dx5 = ops.zeros(n, H, W, cp)
            ops.convlstm1_bwd(xg, ww, bb, dd, None, dx5, cin, F_, x2=(x2, n2))
            dw6, db6 = ops.zeros(3, 3, cin, 4 * F_), ops.zeros(4 * F_)
            ops.convlstm1_bwd(xg, ww, bb, dd, None, None, cin, F_, dw=dw6, dbias=db6, x2=(x2, n2))
            res[name].update(h_x2=h2, dx_x2=dx4, dw_x2=dw4, db_x2=db4, dx5_x2=dx5, dw6_x2=dw6, db6_x2=db6)
            # the gradient of the two high-resolution channels ALONE, added onto a 4-channel buffer that already holds the other
            # branch's gradient (wdg_convlstm1_bwd_dx_from): channels 2, 3 of that buffer stay as they are
            assert ops.convlstm1_dx_from_supported(cin, F_, 3)
            dhi = cv(torch.full((n, H, W, 4), 0.75, dtype=torch.float64))
            ops.convlstm1_bwd(xv, ww, bb, dd, None, dhi, cin, F_, accumulate_dx=True, dx_c0=3)
            dhi0 = cv(torch.full((n, H, W, 4), 0.75, dtype=torch.float64))
            ops.convlstm1_bwd(xv, ww, bb, dd, None, dhi0, cin, F_, accumulate_dx=False, dx_c0=3)
            res[name].update(dhi=dhi, dhi0=dhi0)
    for k in res["ref"]:
        assert rel_err(res["hip"][k], res["ref"][k]) < TOL, k
    if cin == 5:
        assert float((res["hip"]["dhi"][..., 2:] - 0.75).abs().max()) == 0.0 and float((res["hip"]["dhi0"][..., 2:] - 0.75).abs().max()) == 0.0
        assert rel_err(res["hip"]["dhi"][..., :2] - 0.75, res["ref"]["dx2"][..., 3:5]) < 10 * TOL
        assert rel_err(res["hip"]["h_x2"], res["hip"]["h"]) < 1e-6 and rel_err(res["hip"]["dx5_x2"], res["hip"]["dx2"]) < 1e-6
    if Fp > F_:
        assert float(res["hip"]["h"][..., F_:].abs().max()) == 0.0


@pytest.mark.parametrize("cin,n,H,W", [(2, 3, 37, 45), (2, 2, 33, 70), (2, 1, 8, 32)])
def test_fused_conv_layernorm(cin, n, H, W, hip_ops, ref_ops):
    """convln.hip: conv3x3 + bias + LeakyReLU + LayerNorm forward, and its backward (LN', LeakyReLU', dx,
    dgamma/dbeta/dbias) against conv_fwd + ln_fwd / ln_bwd + conv_dgrad of the oracle backend; z and dz are
    channel slices of a wider concat buffer."""
    gen = torch.Generator().manual_seed(17)
    dev = hip_ops.device
    cp = (cin + 3) // 4 * 4
    x = torch.zeros(n, H, W, cp, dtype=torch.float64)
    x[..., :cin] = torch.randn(n, H, W, cin, generator=gen, dtype=torch.float64)
    w = torch.randn(3, 3, cin, 16, generator=gen, dtype=torch.float64) * 0.3
    bias = torch.randn(16, generator=gen, dtype=torch.float64) * 0.3
    gamma = torch.rand(16, generator=gen, dtype=torch.float64) + 0.5
    beta = torch.randn(16, generator=gen, dtype=torch.float64)
    dcat = torch.randn(n, H, W, 32, generator=gen, dtype=torch.float64)
    res = {}
    for name, ops, cv in (("ref", ref_ops, lambda t: t.clone()), ("hip", hip_ops, lambda t: t.float().to(dev))):
        xx, ww, bb, ga, be, dc = cv(x), cv(w).contiguous(), cv(bias), cv(gamma), cv(beta), cv(dcat)
        y, cat, mr = ops.empty(n, H, W, 16), ops.zeros(n, H, W, 32), ops.empty(n * H * W, 2)
        ops.convln_fwd(xx, ww, bb, ga, be, 1e-3, 0.2, y, cat[..., 16:], mr)
        dpre, dx = ops.empty(n, H, W, 16), ops.zeros(n, H, W, cp)
        dg, db, dbias = (cv(torch.ones(16, dtype=torch.float64)) for _ in range(3))
        ops.convln_bwd(dc[..., 16:], y, mr, ww, ga, 0.2, dpre, dx, dg, db, dbias)
        dx2 = ops.zeros(n, H, W, cp)
        dpre2 = ops.empty(n, H, W, 16)
        ops.convln_bwd(dc[..., 16:], y, mr, ww, ga, 0.2, dpre2, dx2, None, None, None)
        # z-only forward + the backward that recomputes y / statistics from x, with the kernel gradient fused in
        cat3 = ops.zeros(n, H, W, 32)
        ops.convln_fwd(xx, ww, bb, ga, be, 1e-3, 0.2, None, cat3[..., 16:], None)
        dx3 = ops.zeros(n, H, W, cp)
        dg3, db3, dbias3 = (cv(torch.ones(16, dtype=torch.float64)) for _ in range(3))
        dw3 = cv(torch.full((3, 3, cin, 16), 0.25, dtype=torch.float64)).contiguous()
        ops.convln_bwd_x(dc[..., 16:], xx, ww, bb, ga, 1e-3, 0.2, dx3, dg3, db3, dbias3, dw3)
        dx4 = ops.zeros(n, H, W, cp)
        ops.convln_bwd_x(dc[..., 16:], xx, ww, bb, ga, 1e-3, 0.2, dx4, None, None, None, None)
        dw5 = ops.zeros(3, 3, cin, 16)
        dg5, db5, dbias5 = ops.zeros(16), ops.zeros(16), ops.zeros(16)
        ops.convln_bwd_x(dc[..., 16:], xx, ww, bb, ga, 1e-3, 0.2, None, dg5, db5, dbias5, dw5)
        res[name] = dict(y=y, cat=cat, mr=mr, dpre=dpre, dx=dx, dg=dg, db=db, dbias=dbias, dx2=dx2, dpre2=dpre2,
                         cat3=cat3, dx3=dx3, dg3=dg3, db3=db3, dbias3=dbias3, dw3=dw3, dx4=dx4, dw5=dw5, dg5=dg5,
                         db5=db5, dbias5=dbias5)
    for k in res["ref"]:
        assert rel_err(res["hip"][k], res["ref"][k]) < TOL, k
    assert rel_err(res["hip"]["dx3"], res["hip"]["dx"]) < TOL and rel_err(res["hip"]["cat3"], res["hip"]["cat"]) == 0.0


def test_sn_power_iter(hip_ops, ref_ops):
    gen = torch.Generator().manual_seed(1)
    for rows, cols in [(8 * 8 * 23, 128), (7 * 7 * 256, 512), (18, 16), (2 * 2 * 32, 192)]:
        w = torch.randn(rows, cols, generator=gen, dtype=torch.float64) * 0.05
        u = torch.randn(cols, generator=gen, dtype=torch.float64) * 0.02
        w_g, u_g = w.float().to(hip_ops.device), u.float().to(hip_ops.device)
        ref_ops.sn_power_iter(w, u)
        hip_ops.sn_power_iter(w_g, u_g)
        assert rel_err(w_g, w) < TOL and rel_err(u_g, u) < TOL, (rows, cols)


def test_prep_batch_equals_layerwise(hip_ops):
    """The batched network preparation (all SN power iterations + repacks in one launch per stage) is
    bit-identical to the layer-by-layer entry points, for SN and plain layers, Cout % 4 != 0 included."""
    gen = torch.Generator().manual_seed(3)
    shapes = [(8, 8, 23, 128, True), (3, 3, 16, 2, False), (7, 7, 64, 128, True), (2, 2, 32, 192, True),
              (3, 3, 5, 64, False), (3, 3, 2, 16, True)]
    dev = hip_ops.device
    ws = [(torch.randn(kh, kw, ci, co, generator=gen) * 0.05).to(dev) for kh, kw, ci, co, _ in shapes]
    us = [(torch.randn(co, generator=gen) * 0.02).to(dev) if sn else None for _, _, _, co, sn in shapes]
    wa, ua = [w.clone() for w in ws], [None if u is None else u.clone() for u in us]
    pka = [hip_ops.pack_weights(w) for w in wa]
    pkb = [hip_ops.pack_weights(w) for w in ws]
    batch = hip_ops.make_prep_batch(list(zip(pkb, us)))
    for _ in range(2):
        for pk, u in zip(pka, ua):          # layer by layer
            if u is not None:
                hip_ops.sn_power_iter(pk.w.view(-1, pk.cout), u)
                pk.refresh()
        batch.run(sn=True, pack_all=False)
    for pk_a, pk_b, u_a, u_b in zip(pka, pkb, ua, us):
        assert torch.equal(pk_a.w, pk_b.w) and torch.equal(pk_a.wF, pk_b.wF) and torch.equal(pk_a.wD, pk_b.wD)
        if u_a is not None:
            assert torch.equal(u_a, u_b)
    # optimizer-step path: masters change, everything is repacked, no SN
    for pk_a, pk_b in zip(pka, pkb):
        pk_a.w.mul_(1.5)
        pk_b.w.mul_(1.5)
        pk_a.refresh()
    batch.run(sn=False, pack_all=True)
    for pk_a, pk_b in zip(pka, pkb):
        assert torch.equal(pk_a.wF, pk_b.wF) and torch.equal(pk_a.wD, pk_b.wD)


@pytest.mark.parametrize("C,P", [(128, 5000), (64, 777), (32, 4096), (16, 70000)])
def test_batchnorm(C, P, hip_ops, ref_ops):
    gen = torch.Generator().manual_seed(2)
    dev = hip_ops.device
    y = torch.randn(P, C, generator=gen, dtype=torch.float64) * 1.5 + 0.3
    dz = torch.randn(P, C, generator=gen, dtype=torch.float64)
    gamma = torch.rand(C, generator=gen, dtype=torch.float64) + 0.5
    beta = torch.randn(C, generator=gen, dtype=torch.float64)
    res = {}
    for name, ops, cv in (("ref", ref_ops, lambda t: t.clone()), ("hip", hip_ops, lambda t: t.float().to(dev))):
        yy, dzz, ga, be = cv(y), cv(dz), cv(gamma), cv(beta)
        stats = ops.zeros(3, 2 * C, dtype=torch.float64)     # replica slabs: the standalone pass fills slab 0
        mm, mv = cv(torch.zeros(C, dtype=torch.float64)), cv(torch.ones(C, dtype=torch.float64))
        ss, saved = ops.empty(2 * C), ops.empty(2 * C)
        ops.bn_stats(yy, stats[0])
        ops.bn_finalize_train(stats, P, ga, be, mm, mv, 0.99, 1e-3, ss, saved)
        z = ops.empty(P, C)
        ops.bn_apply(yy, ss, z)
        red = ops.zeros(2 * C, dtype=torch.float64)
        ops.bn_bwd_reduce(dzz, yy, saved, red)
        dpre = ops.empty(P, C)
        dg, db, dbias = cv(torch.zeros(C, dtype=torch.float64)), cv(torch.zeros(C, dtype=torch.float64)), cv(torch.zeros(C, dtype=torch.float64))
        ops.bn_bwd_apply(dzz, yy, saved, ga, red, red, P, 0.2, dpre, dg, db, dbias)
        ss2 = ops.empty(2 * C)
        ops.bn_finalize_infer(ga, be, mm, mv, 1e-3, ss2)
        res[name] = dict(z=z, mm=mm, mv=mv, dpre=dpre, dg=dg, db=db, dbias=dbias, ss2=ss2)
    # independent check of the oracle itself against torch's batch_norm
    zz = torch.nn.functional.batch_norm(y.t()[None], None, None, gamma, beta, True, 0.0, 1e-3)[0].t()
    assert rel_err(res["ref"]["z"], zz) < 1e-12
    for k in res["ref"]:
        assert rel_err(res["hip"][k], res["ref"][k]) < TOL, k


BN_HOOK_CASES = [
    # name, n, H, W, cin, cout, k, s, p, direction
    ("g0_8x8s2_cin23", 3, 64, 64, 23, 128, 8, 2, 3, "fwd"),          # 128x128 tile
    ("g2_4x4s2", 3, 32, 32, 128, 128, 4, 2, 1, "fwd"),
    ("g5_3x3_128to64", 3, 48, 40, 128, 64, 3, 1, 1, "fwd"),          # 128x64 tile / 64x64
    ("g7_convT2x2", 2, 32, 32, 32, 192, 2, 2, 0, "dgrad"),           # four dgrad phases, 256x32 tile
    ("splitk_small_m", 8, 8, 8, 256, 512, 7, 3, 1, "fwd"),           # split-K route: standalone pass behind the conv
    ("halo_route", 4, 128, 128, 16, 16, 3, 1, 1, "fwd"),             # halo-tile kernel route
    ("ragged_cout", 2, 20, 24, 16, 40, 3, 1, 1, "fwd"),
]


@pytest.mark.parametrize("case", BN_HOOK_CASES, ids=[c[0] for c in BN_HOOK_CASES])
def test_conv_batchnorm_hooks(case, hip_ops, ref_ops):
    """wdg_conv_fwd_bn / wdg_conv_dgrad_bn: the BatchNormalization that follows a conv (models.py:33-34 ff.) folded into
    the launch — training: per-channel sum / sum of squares of the activated output land in the replica slabs;
    inference: the [scale | shift] affine is applied by the epilogue.  Same results on every route."""
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    name, n, H, W, cin, cout, k, s_, p_, direction = case
    gen = torch.Generator().manual_seed(17)
    dev = hip_ops.device
    Ho, Wo = (H + 2 * p_ - k) // s_ + 1, (W + 2 * p_ - k) // s_ + 1
    w = torch.randn(k, k, cin, cout, generator=gen, dtype=torch.float64) / np.sqrt(k * k * cin)
    cinp, coutp = (cin + 3) // 4 * 4, (cout + 3) // 4 * 4
    if direction == "fwd":
        src = torch.zeros(n, H, W, cinp, dtype=torch.float64)
        src[..., :cin] = torch.randn(n, H, W, cin, generator=gen, dtype=torch.float64)
        C, oshape = cout, (n, Ho, Wo, coutp)
    else:
        src = torch.zeros(n, Ho, Wo, coutp, dtype=torch.float64)
        src[..., :cout] = torch.randn(n, Ho, Wo, cout, generator=gen, dtype=torch.float64)
        C, oshape = cin, (n, H, W, cinp)
    bias = torch.randn(C, generator=gen, dtype=torch.float64) * 0.3
    aff = torch.cat([torch.rand(C, generator=gen, dtype=torch.float64) + 0.5, torch.randn(C, generator=gen, dtype=torch.float64)])
    res = {}
    for tag, ops, cv, G in (("ref", ref_ops, lambda t: t.clone(), RG), ("hip", hip_ops, lambda t: t.float().to(dev).contiguous(), ConvGeom)):
        pk = ops.pack_weights(cv(w))
        g = G(k, k, s_, p_)
        call = (lambda out, **kw: ops.conv_fwd(cv(src), pk, cv(bias), out, g, act=True, slope=0.2, **kw)) if direction == "fwd" else \
               (lambda out, **kw: ops.conv_dgrad(cv(src), pk, out, g, bias=cv(bias), act=True, slope=0.2, **kw))
        plain, y1, y2 = ops.zeros(*oshape), ops.zeros(*oshape), ops.zeros(*oshape)
        call(plain)
        stats = ops.zeros(5, 2 * C, dtype=torch.float64)
        call(y1, bn_stats=stats)
        call(y2, bn_affine=cv(aff))
        res[tag] = dict(plain=plain, y1=y1, stats=stats.sum(0), y2=y2)
    r, h = res["ref"], res["hip"]
    assert torch.equal(h["y1"], h["plain"])                              # the statistics hook does not touch the output
    assert rel_err(h["plain"], r["plain"]) < TOL
    assert rel_err(h["stats"][:C], r["stats"][:C]) < TOL and rel_err(h["stats"][C:], r["stats"][C:]) < TOL
    # oracle consistency: statistics are those of the written tensor
    assert rel_err(r["stats"][:C], r["plain"][..., :C].sum((0, 1, 2))) < 1e-12
    assert rel_err(h["y2"], r["y2"]) < TOL
    assert rel_err(r["y2"][..., :C], r["plain"][..., :C] * aff[:C] + aff[C:]) < 1e-12
    assert float(h["y2"][..., C:].abs().max() if oshape[3] > C else 0.0) == 0.0          # pad channels stay zero


LN_CASES = [
    # name, n, H, W, cin, cout, k, s, p
    ("d_block1_epilogue", 3, 96, 96, 32, 64, 7, 3, 1),          # 128x64 tile spans all 64 channels: norm in the igemm epilogue
    ("d_block2_two_tiles", 3, 30, 30, 64, 128, 7, 3, 1),         # 64-wide tiles, 128 channels: standalone pass behind the conv
    ("d_block3_splitk", 8, 27, 27, 128, 256, 7, 3, 1),           # split-K: norm in the second stage
    ("d_block4_splitk", 8, 8, 8, 256, 512, 7, 3, 1),
    ("tail_3x3s2", 4, 3, 3, 256, 512, 3, 2, 0),
    ("encoder_5x5s3", 3, 40, 40, 4, 8, 5, 3, 1),
    ("cout128_one_tile", 2, 64, 64, 128, 128, 3, 1, 1),          # 128x128 tile, epilogue
    ("d_conv_b_16to16_halo", 3, 40, 52, 16, 16, 3, 1, 1),        # thin full-resolution layer: norm in the halo-tile kernel's epilogue (ragged tiles)
    ("d_conv_b_16to16_halo_256", 2, 256, 256, 16, 16, 3, 1, 1),
]


@pytest.mark.parametrize("case", LN_CASES, ids=[c[0] for c in LN_CASES])
def test_conv_layernorm_fused(case, hip_ops, ref_ops):
    """wdg_conv_fwd_ln: conv -> bias -> LeakyReLU -> LayerNormalization (models.py:113-116 ff.) with the norm in the
    epilogue that owns complete rows; y, z and (mean, rstd) against the two reference layers applied one after the other."""
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    name, n, H, W, cin, cout, k, s_, p_ = case
    gen = torch.Generator().manual_seed(31)
    dev = hip_ops.device
    Ho, Wo = (H + 2 * p_ - k) // s_ + 1, (W + 2 * p_ - k) // s_ + 1
    x = torch.randn(n, H, W, cin, generator=gen, dtype=torch.float64)
    w = torch.randn(k, k, cin, cout, generator=gen, dtype=torch.float64) / np.sqrt(k * k * cin)
    bias = torch.randn(cout, generator=gen, dtype=torch.float64) * 0.3
    gamma = torch.rand(cout, generator=gen, dtype=torch.float64) + 0.5
    beta = torch.randn(cout, generator=gen, dtype=torch.float64)
    res = {}
    for tag, ops, cv, G in (("ref", ref_ops, lambda t: t.clone(), RG), ("hip", hip_ops, lambda t: t.float().to(dev).contiguous(), ConvGeom)):
        pk, g = ops.pack_weights(cv(w)), G(k, k, s_, p_)
        y = ops.zeros(n, Ho, Wo, cout)
        # the thin-layer cases write z into the second half of a wider buffer, as the discriminator does (models.py:102-108)
        zbuf = ops.zeros(n, Ho, Wo, 2 * cout) if "halo" in name else ops.zeros(n, Ho, Wo, cout)
        z = zbuf[..., cout:] if "halo" in name else zbuf
        mr = ops.empty(n * Ho * Wo, 2)
        ops.conv_fwd_ln(cv(x), pk, cv(bias), y, z, g, cv(gamma), cv(beta), 1e-3, mr, act=True, slope=0.2)
        if "halo" in name:
            assert float(zbuf[..., :cout].abs().max()) == 0.0          # the other half of the concatenation is untouched
            z = z.contiguous()
        y2 = ops.zeros(n, Ho, Wo, cout)
        ops.conv_fwd(cv(x), pk, cv(bias), y2, g, act=True, slope=0.2)
        res[tag] = dict(y=y, z=z, mr=mr, y2=y2)
    r, h = res["ref"], res["hip"]
    assert rel_err(h["y"], r["y"]) < TOL and rel_err(h["y"], h["y2"]) < 1e-6
    assert rel_err(h["z"], r["z"]) < TOL
    assert rel_err(h["mr"][:, 0], r["mr"][:, 0]) < TOL and rel_err(h["mr"][:, 1], r["mr"][:, 1]) < TOL
    zz = torch.nn.functional.layer_norm(r["y"], (cout,), gamma, beta, 1e-3)
    assert rel_err(r["z"], zz) < 1e-12


LNB_CASES = [
    # name, n, H, W, cin (= channels of dx), cout, k, s, p, c0, C   (the LayerNorm group is channels [c0, c0 + C) of dx)
    ("d_block0_group16_of_32", 3, 96, 96, 32, 64, 7, 3, 1, 16, 16),     # 256x32 tile, 9 phases: the norm of the low + high branch (models.py:105)
    ("d_block0_full_map", 2, 256, 256, 32, 64, 7, 3, 1, 16, 16),         # dgrad_patch_s3.hip: 11 x 11 tiles of 8 x 8 bases, ragged last tiles
    ("d_block0_same_pad_all32", 2, 50, 77, 32, 64, 7, 3, 3, 0, 32),      # 'same' padding, all 32 channels one group, H != W
    ("d_block0_valid_group8", 2, 64, 40, 32, 64, 7, 3, 0, 8, 8),         # no padding, a group inside one channel tile
    ("d_block1_64ch", 3, 31, 31, 64, 128, 7, 3, 1, 0, 64),               # 128x64 tile on 4 x 1 waves
    ("d_block1_64ch_big", 4, 84, 84, 64, 128, 7, 3, 1, 0, 64),
    ("d_block2_128ch_tile64x128", 8, 27, 27, 128, 256, 7, 3, 1, 0, 128),  # 64x64 tiling -> the 64x128 tile of the fused route
    ("d_block3_splitk_256ch", 8, 8, 8, 256, 512, 7, 3, 1, 0, 256),        # split-K: the norm's backward in the second stage
    ("tail_3x3s2_splitk", 4, 5, 5, 256, 512, 3, 2, 0, 0, 256),
    ("group8_of_16_thin", 3, 40, 40, 16, 32, 7, 3, 1, 8, 8),             # test-width discriminator (feature_channels = 8)
    ("halo_route_fallback", 2, 128, 128, 16, 16, 3, 1, 1, 0, 16),         # halo-tile data gradient: the two-launch route
]


@pytest.mark.parametrize("par", [True, False], ids=["param_grads", "input_grad_only"])
@pytest.mark.parametrize("case", LNB_CASES, ids=[c[0] for c in LNB_CASES])
def test_conv_dgrad_layernorm_backward_fused(case, par, hip_ops, ref_ops):
    """wdg_conv_dgrad_lnbwd: the data gradient of a convolution chained with the LayerNorm + LeakyReLU backward of the layer that
    produced (a channel group of) its input — dx, dgamma, dbeta, dbias (accumulated onto non-zero starting values) against
    conv_dgrad followed by ln_bwd of the oracle; the scratch is left zeroed; the untouched channels are the plain data gradient."""
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    name, n, H, W, cin, cout, k, s_, p_, c0, C = case
    gen = torch.Generator().manual_seed(41)
    dev = hip_ops.device
    Ho, Wo = (H + 2 * p_ - k) // s_ + 1, (W + 2 * p_ - k) // s_ + 1
    dy = torch.randn(n, Ho, Wo, cout, generator=gen, dtype=torch.float64)
    w = torch.randn(k, k, cin, cout, generator=gen, dtype=torch.float64) / np.sqrt(k * k * cout)
    y = torch.randn(n, H, W, C, generator=gen, dtype=torch.float64)          # the norm's input (post-LeakyReLU activation)
    gamma = torch.rand(C, generator=gen, dtype=torch.float64) + 0.5
    mean = y.mean(-1).reshape(-1)
    rstd = 1.0 / torch.sqrt(y.var(-1, unbiased=False).reshape(-1) + 1e-3)
    mr = torch.stack([mean, rstd], 1).contiguous()
    start = torch.randn(3, C, generator=gen, dtype=torch.float64)
    res = {}
    for tag, ops, cv, G in (("ref", ref_ops, lambda t: t.clone(), RG), ("hip", hip_ops, lambda t: t.float().to(dev).contiguous(), ConvGeom)):
        pk, g = ops.pack_weights(cv(w)), G(k, k, s_, p_)
        dx = ops.zeros(n, H, W, cin)
        dg, db, dbias = (cv(start[0]), cv(start[1]), cv(start[2])) if par else (None, None, None)
        ws = ops.lnbwd_scratch(C) if par else None
        ops.conv_dgrad_lnbwd(cv(dy), pk, dx, g, cv(y), cv(mr), cv(gamma), c0, C, 0.2, dg, db, dbias, ws)
        plain = ops.zeros(n, H, W, cin)
        ops.conv_dgrad(cv(dy), pk, plain, g)
        res[tag] = dict(dx=dx, dg=dg, db=db, dbias=dbias, plain=plain, ws=ws)
    r, h = res["ref"], res["hip"]
    assert rel_err(h["dx"], r["dx"]) < TOL
    if c0 > 0:
        assert rel_err(h["dx"][..., :c0], h["plain"][..., :c0]) < 1e-6
    assert rel_err(h["dx"][..., c0:c0 + C], h["plain"][..., c0:c0 + C]) > 1e-2     # (the group did go through the norm's backward)
    if par:
        for key in ("dg", "db", "dbias"):
            assert rel_err(h[key], r[key]) < 5 * TOL, key                           # (sums over up to 10^5 pixels in fp32 atomics)
        assert float(h["ws"].abs().max()) == 0.0
        # a second call accumulates again from the zeroed scratch
        ops, cv = hip_ops, (lambda t: t.float().to(dev).contiguous())
        dx2 = ops.zeros(n, H, W, cin)
        ops.conv_dgrad_lnbwd(cv(dy), ops.pack_weights(cv(w)), dx2, ConvGeom(k, k, s_, p_), cv(y), cv(mr), cv(gamma), c0, C, 0.2,
                             h["dg"], h["db"], h["dbias"], h["ws"])
        assert rel_err(h["dg"], 2 * r["dg"] - start[0]) < 5 * TOL and rel_err(dx2, r["dx"]) < TOL
    if name.startswith("d_block0"):
        # the result does not depend on the route: the patch kernel (default) against the implicit-GEMM epilogue
        ops, cv = hip_ops, (lambda t: t.float().to(dev).contiguous())
        assert ops.lib.wdg_set_tuning(b"dgrad_s3", 0) == 0
        try:
            dx3 = ops.zeros(n, H, W, cin)
            ops.conv_dgrad_lnbwd(cv(dy), ops.pack_weights(cv(w)), dx3, ConvGeom(k, k, s_, p_), cv(y), cv(mr), cv(gamma), c0, C, 0.2,
                                 None, None, None, None)
        finally:
            ops.lib.wdg_set_tuning(b"dgrad_s3", 1)
        assert rel_err(dx3, h["dx"]) < 1e-5 and not torch.equal(dx3, h["dx"])      # (different reduction order: close, not identical)


def test_conv_dgrad_layernorm_backward_routes_and_repeatability(hip_ops):
    """wdg_conv_dgrad_lnbwd_route names the kernels a call launches: the 7x7 stride-3 32 -> 64 layer goes to the patch kernel
    (csrc/dgrad_patch_s3.hip), whose parameter sums — plain stores + a fixed-order finish kernel, no atomics — are the same bits on
    every run; the other fused layers stay in the implicit-GEMM epilogue; with the patch kernel switched off so does that layer."""
    from downscaling.engine.hipops import ConvGeom
    ops, dev = hip_ops, hip_ops.device
    gen = torch.Generator().manual_seed(5)

    def make(n, H, cin, cout, C):
        Ho = (H + 2 - 7) // 3 + 1
        dy = torch.randn(n, Ho, Ho, cout, generator=gen).to(dev)
        pk = ops.pack_weights((torch.randn(7, 7, cin, cout, generator=gen) / 50).to(dev))
        y = torch.randn(n, H, H, C, generator=gen).to(dev)
        mr = torch.stack([y.mean(-1).reshape(-1), 1.0 / torch.sqrt(y.var(-1, unbiased=False).reshape(-1) + 1e-3)], 1).contiguous()
        return dy, pk, ops.zeros(n, H, H, cin), y, mr, (torch.rand(C, generator=gen) + 0.5).to(dev)

    g = ConvGeom(7, 7, 3, 1)
    dy, pk, dx, y, mr, gamma = make(3, 96, 32, 64, 16)
    assert ops.conv_dgrad_lnbwd_route(dy, pk, dx, g, y, 16, 16) == 2
    runs = []
    for _ in range(3):
        grads = [torch.zeros(16, device=dev) for _ in range(3)]
        ops.conv_dgrad_lnbwd(dy, pk, dx, g, y, mr, gamma, 16, 16, 0.2, *grads, ops.lnbwd_scratch(16))
        runs.append([t.clone() for t in grads] + [dx.clone()])
    for r in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(r, runs[0]))
    assert ops.lib.wdg_set_tuning(b"dgrad_s3", 0) == 0
    try:
        assert ops.conv_dgrad_lnbwd_route(dy, pk, dx, g, y, 16, 16) == 1
    finally:
        ops.lib.wdg_set_tuning(b"dgrad_s3", 1)
    dy2, pk2, dx2, y2, _, _ = make(2, 31, 64, 128, 64)
    assert ops.conv_dgrad_lnbwd_route(dy2, pk2, dx2, g, y2, 0, 64) == 1


@pytest.mark.parametrize("B,T,npix,C,par", [(4, 1, 4, 512, True), (3, 2, 9, 64, True), (2, 3, 1, 256, False), (32, 1, 4, 512, True)])
def test_dense_head_backward_through_layernorm(B, T, npix, C, par, hip_ops, ref_ops):
    """wdg_dense_gap_bwd_ln: Dense(1) + GlobalAveragePooling backward (models.py:137-140) chained with the backward of the
    LayerNormalization that produced the head's input, against dense_gap_bwd + ln_bwd of the oracle."""
    gen = torch.Generator().manual_seed(43)
    dev = hip_ops.device
    rows, K = B * T, npix * C
    y = torch.randn(rows * npix, C, generator=gen, dtype=torch.float64)
    gamma = torch.rand(C, generator=gen, dtype=torch.float64) + 0.5
    beta = torch.randn(C, generator=gen, dtype=torch.float64)
    mean, var = y.mean(1), y.var(1, unbiased=False)
    mr = torch.stack([mean, 1.0 / torch.sqrt(var + 1e-3)], 1).contiguous()
    x = (((y - mean[:, None]) * mr[:, 1:2]) * gamma + beta).reshape(rows, K).contiguous()      # the norm's output = the head's input
    w = torch.randn(K, generator=gen, dtype=torch.float64) / np.sqrt(K)
    dscore = torch.randn(B, generator=gen, dtype=torch.float64)
    start = torch.randn(3, C, generator=gen, dtype=torch.float64)
    res = {}
    for tag, ops, cv in (("ref", ref_ops, lambda t: t.clone()), ("hip", hip_ops, lambda t: t.float().to(dev).contiguous())):
        dx, dw, dbd = ops.zeros(rows, K), ops.zeros(K), ops.zeros(1)
        dg, db, dbias = (cv(start[0]), cv(start[1]), cv(start[2])) if par else (None, None, None)
        ops.dense_gap_bwd_ln(cv(x), cv(w), cv(dscore), dx, dw if par else None, dbd if par else None, B, T, cv(y), cv(mr), cv(gamma), C,
                             0.2, dg, db, dbias, ops.lnbwd_scratch(C) if par else None)
        res[tag] = dict(dx=dx, dw=dw, dbd=dbd, dg=dg, db=db, dbias=dbias)
    r, h = res["ref"], res["hip"]
    assert rel_err(h["dx"], r["dx"]) < TOL
    if par:
        for key in ("dw", "dbd", "dg", "db", "dbias"):
            assert rel_err(h[key], r[key]) < 5 * TOL, key


def test_upconv_batchnorm_hooks(hip_ops, ref_ops):
    """The fused upsample + 5x5 transposed-conv block (models.py:62-64) as a BatchNormalization producer: statistics /
    inference affine in the column-form gather kernel."""
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    gen = torch.Generator().manual_seed(23)
    dev = hip_ops.device
    x = torch.randn(3, 36, 44, 160, generator=gen, dtype=torch.float64)
    w = torch.randn(5, 5, 16, 160, generator=gen, dtype=torch.float64) * 0.03
    b = torch.randn(16, generator=gen, dtype=torch.float64) * 0.2
    aff = torch.cat([torch.rand(16, generator=gen, dtype=torch.float64) + 0.5, torch.randn(16, generator=gen, dtype=torch.float64)])
    out = {}
    for tag, ops, cv, G in (("ref", ref_ops, lambda t: t.clone(), RG), ("hip", hip_ops, lambda t: t.float().to(dev).contiguous(), ConvGeom)):
        pk, g = ops.pack_weights(cv(w)), G(5, 5, 1, 2)
        y0, y1, y2 = ops.zeros(3, 72, 88, 16), ops.zeros(3, 72, 88, 16), ops.zeros(3, 72, 88, 16)
        stats = ops.zeros(4, 32, dtype=torch.float64)
        ops.upconv_fwd(cv(x), pk, cv(b), y0, g, act=True)
        ops.upconv_fwd(cv(x), pk, cv(b), y1, g, act=True, bn_stats=stats)
        ops.upconv_fwd(cv(x), pk, cv(b), y2, g, act=True, bn_affine=cv(aff))
        out[tag] = dict(y0=y0, y1=y1, y2=y2, stats=stats.sum(0))
    r, h = out["ref"], out["hip"]
    assert torch.equal(h["y0"], h["y1"])
    for k_ in ("y0", "y2"):
        assert rel_err(h[k_], r[k_]) < TOL, k_
    assert rel_err(h["stats"][:16], r["stats"][:16]) < TOL and rel_err(h["stats"][16:], r["stats"][16:]) < TOL


@pytest.mark.parametrize("C,P", [(16, 5000), (32, 1234), (64, 999), (128, 500), (256, 130), (512, 77)])
def test_layernorm(C, P, hip_ops, ref_ops):
    gen = torch.Generator().manual_seed(3)
    dev = hip_ops.device
    y = torch.randn(P, C, generator=gen, dtype=torch.float64) * 2 + 0.5
    dz = torch.randn(P, C, generator=gen, dtype=torch.float64)
    gamma = torch.rand(C, generator=gen, dtype=torch.float64) + 0.5
    beta = torch.randn(C, generator=gen, dtype=torch.float64)
    res = {}
    for name, ops, cv in (("ref", ref_ops, lambda t: t.clone()), ("hip", hip_ops, lambda t: t.float().to(dev))):
        yy, dzz, ga, be = cv(y), cv(dz), cv(gamma), cv(beta)
        z, mr, dpre = ops.empty(P, C), ops.empty(P, 2), ops.empty(P, C)
        ops.ln_fwd(yy, ga, be, 1e-3, z, mr)
        dg, db, dbias = (cv(torch.zeros(C, dtype=torch.float64)) for _ in range(3))
        ops.ln_bwd(dzz, yy, mr, ga, 0.2, dpre, dg, db, dbias)
        res[name] = dict(z=z, dpre=dpre, dg=dg, db=db, dbias=dbias)
    for k in res["ref"]:
        assert rel_err(res["hip"][k], res["ref"][k]) < TOL, k


@pytest.mark.parametrize("F_,P", [(128, 3000), (2, 5000), (16, 4097)])
def test_lstm_cell(F_, P, hip_ops, ref_ops):
    gen = torch.Generator().manual_seed(4)
    dev = hip_ops.device
    Fp = (F_ + 3) // 4 * 4
    gates = torch.randn(P, 4 * F_, generator=gen, dtype=torch.float64) * 2.5
    cprev = torch.randn(P, F_, generator=gen, dtype=torch.float64)
    dh = torch.randn(P, Fp, generator=gen, dtype=torch.float64)
    dcin = torch.randn(P, F_, generator=gen, dtype=torch.float64)
    for use_prev in (True, False):
        res = {}
        for name, ops, cv in (("ref", ref_ops, lambda t: t.clone()), ("hip", hip_ops, lambda t: t.float().to(dev))):
            g_, cp, dh_, dci = cv(gates), (cv(cprev) if use_prev else None), cv(dh), (cv(dcin) if use_prev else None)
            c, h = ops.empty(P, F_), ops.zeros(P, Fp)
            ops.lstm_fwd(g_, cp, c, h, F_)
            dg = ops.empty(P, 4 * F_)
            dcp = ops.empty(P, F_) if use_prev else None
            ops.lstm_bwd(g_, cp, c, dh_, dci, dg, dcp, F_)
            res[name] = dict(c=c, h=h, dg=dg)
            if use_prev:
                res[name]["dcp"] = dcp
        for k in res["ref"]:
            assert rel_err(res["hip"][k], res["ref"][k]) < TOL, (k, use_prev)


def test_upsample_dense_misc(hip_ops, ref_ops):
    gen = torch.Generator().manual_seed(6)
    dev = hip_ops.device
    cv = lambda t: t.float().to(dev)
    # bilinear x2 forward + adjoint
    x = torch.randn(2, 7, 9, 160, generator=gen, dtype=torch.float64)
    dy = torch.randn(2, 14, 18, 160, generator=gen, dtype=torch.float64)
    y_r, y_g = torch.zeros(2, 14, 18, 160, dtype=torch.float64), hip_ops.empty(2, 14, 18, 160)
    ref_ops.upsample2x_fwd(x, y_r)
    hip_ops.upsample2x_fwd(cv(x), y_g)
    assert rel_err(y_g, y_r) < TOL
    dx_r, dx_g = torch.ones_like(x), cv(torch.ones_like(x))
    ref_ops.upsample2x_bwd(dy, dx_r, accumulate=True)
    hip_ops.upsample2x_bwd(cv(dy), dx_g, accumulate=True)
    assert rel_err(dx_g, dx_r) < TOL
    # dense + GAP (time-major rows)
    B, T, K = 5, 3, 2048
    xr = torch.randn(T * B, K, generator=gen, dtype=torch.float64)
    w = torch.randn(K, generator=gen, dtype=torch.float64) * 0.05
    b = torch.randn(1, generator=gen, dtype=torch.float64)
    ds = torch.randn(B, generator=gen, dtype=torch.float64)
    s_r, s_g = torch.zeros(B, dtype=torch.float64), hip_ops.empty(B)
    ref_ops.dense_gap_fwd(xr, w, b, s_r, B, T)
    hip_ops.dense_gap_fwd(cv(xr), cv(w), cv(b), s_g, B, T)
    assert rel_err(s_g, s_r) < TOL
    dxr, dwr, dbr = torch.zeros_like(xr), torch.ones_like(w), torch.ones(1, dtype=torch.float64)
    dxg, dwg, dbg = hip_ops.empty(T * B, K), cv(torch.ones_like(w)), cv(torch.ones(1, dtype=torch.float64))
    ref_ops.dense_gap_bwd(xr, w, ds, dxr, dwr, dbr, B, T)
    hip_ops.dense_gap_bwd(cv(xr), cv(w), cv(ds), dxg, dwg, dbg, B, T)
    assert rel_err(dxg, dxr) < TOL and rel_err(dwg, dwr) < TOL and rel_err(dbg, dbr) < TOL
    # channel pack with image permutation, colsum, lerp, sumsq, segment mean-square, adam
    src = torch.randn(4, 6, 6, 3, generator=gen, dtype=torch.float64)
    dst_r, dst_g = torch.zeros(4, 6, 6, 8, dtype=torch.float64), hip_ops.zeros(4, 6, 6, 8)
    ref_ops.copy_channels(src, dst_r[..., 0:3])
    hip_ops.copy_channels(cv(src), dst_g[..., 0:3])
    ref_ops.copy_channels(src[..., :2], dst_r[..., 3:5], accumulate=True)
    hip_ops.copy_channels(cv(src)[..., :2], dst_g[..., 3:5], accumulate=True)
    assert rel_err(dst_g, dst_r) < 1e-7
    bt = torch.randn(2, 3, 5, 5, 4, generator=gen, dtype=torch.float64)  # [B,T,...] -> time-major
    tm_r, tm_g = torch.zeros(3, 2, 5, 5, 4, dtype=torch.float64), hip_ops.zeros(3, 2, 5, 5, 4)
    for t in range(3):
        ref_ops.copy_channels(bt[:, t], tm_r[t])
        hip_ops.copy_channels(cv(bt)[:, t], tm_g[t])
    assert rel_err(tm_g, bt.transpose(0, 1)) < 1e-7
    xx = torch.randn(3000, 512, generator=gen, dtype=torch.float64)
    o_r, o_g = torch.ones(512, dtype=torch.float64), cv(torch.ones(512, dtype=torch.float64))
    ref_ops.colsum(xx, o_r, accumulate=True)
    hip_ops.colsum(cv(xx), o_g, accumulate=True)
    assert rel_err(o_g, o_r) < TOL
    Tt, Bb, ppi = 2, 3, 50
    a, bb = torch.randn(Tt * Bb * ppi, 4, generator=gen, dtype=torch.float64), torch.randn(Tt * Bb * ppi, 4, generator=gen, dtype=torch.float64)
    eps = torch.rand(Bb, generator=gen, dtype=torch.float64)
    l_r, l_g = torch.zeros_like(a), hip_ops.empty(*a.shape)
    ref_ops.lerp_batch(a, bb, eps, l_r, ppi, Bb)
    hip_ops.lerp_batch(cv(a), cv(bb), cv(eps), l_g, ppi, Bb)
    assert rel_err(l_g, l_r) < 1e-6
    q_r, q_g = torch.zeros(Bb, 4, dtype=torch.float64), hip_ops.empty(Bb, 4)
    ref_ops.sumsq_batch_ch(a, ppi, Tt, Bb, q_r)
    hip_ops.sumsq_batch_ch(cv(a), ppi, Tt, Bb, q_g)
    assert rel_err(q_g, q_r) < TOL
    flat = torch.randn(10000, generator=gen, dtype=torch.float64)
    off = torch.tensor([0, 10, 12, 5000, 5000, 10000], dtype=torch.int64)  # {begin, end} pairs
    m_r, m_g = torch.zeros(3, dtype=torch.float64), hip_ops.empty(3)
    ref_ops.segment_meansq(flat, off, m_r)
    hip_ops.segment_meansq(cv(flat), off.to(dev), m_g)
    assert rel_err(m_g, m_r) < TOL
    p, gr = torch.randn(5000, generator=gen, dtype=torch.float64), torch.randn(5000, generator=gen, dtype=torch.float64) * 0.01
    m, v = torch.zeros(5000, dtype=torch.float64), torch.zeros(5000, dtype=torch.float64)
    pg, mg, vg, gg = cv(p), cv(m), cv(v), cv(gr)
    for step in (1, 2, 3):
        lr_t = 1e-4 * np.sqrt(1 - 0.9 ** step) / (1 - 0.5 ** step)
        ref_ops.adam_tf(p, gr, m, v, lr_t, 0.5, 0.9, 0.1, 0.5)
        hip_ops.adam_tf(pg, gg, mg, vg, lr_t, 0.5, 0.9, 0.1, 0.5)
    assert rel_err(pg, p) < 1e-6 and rel_err(mg, m) < 1e-5 and rel_err(vg, v) < 1e-5


def test_philox_noise(hip_ops, ref_ops):
    """Bitstream parity with the numpy Philox4x32-10 restatement; logf/cosf ulps only."""
    dev = hip_ops.device
    out_r, out_g = torch.zeros(1001, 20, dtype=torch.float64), hip_ops.zeros(1001, 24)
    ref_ops.philox_normal(out_r, 1234567, 77, 0.1)
    hip_ops.philox_normal(out_g[:, :20], 1234567, 77, 0.1)
    assert float((out_g[:, :20].double().cpu() - out_r).abs().max()) < 1e-6
    assert float(out_g[:, 20:].abs().max()) == 0.0
    assert abs(float(out_g[:, :20].std()) - 0.1) < 2e-3
    u_r, u_g = torch.zeros(37, dtype=torch.float64), hip_ops.empty(37)
    ref_ops.philox_uniform(u_r, 99, 5)
    hip_ops.philox_uniform(u_g, 99, 5)
    assert float((u_g.double().cpu() - u_r).abs().max()) == 0.0


def test_structured_noise_generator(hip_ops):
    """NoiseGenerator (data_generator.py:296-316) on the Philox kernel: shape, scale, and the reference's layout (the
    time-varying channel is constant over (x, y); every draw fills a run of consecutive elements)."""
    from downscaling.data.data_generator import NoiseGenerator
    from downscaling.engine import runtime
    runtime.set_ops(hip_ops)
    gen = NoiseGenerator((4, 6, 8, 10, 4), std=0.5, random_seed=11)
    n = gen()
    assert tuple(n.shape) == (4, 6, 8, 10, 4) and n.is_cuda
    tv = n[..., 0]
    assert float((tv - tv[:, :, :1, :1]).abs().max()) == 0.0
    lon = n[..., 1].reshape(4, -1)
    runs = lon.view(4, 8, 60)                      # bs x draws x run length t*y
    assert float((runs - runs[:, :, :1]).abs().max()) == 0.0
    assert 0.3 < float(n[..., 3].std()) < 0.7
    n2 = NoiseGenerator((4, 6, 8, 10, 4), std=0.5, random_seed=11)(bs=2)
    assert tuple(n2.shape) == (2, 6, 8, 10, 4)


@pytest.mark.parametrize("cin,F,B,T,H,W", [(5, 16, 2, 3, 24, 32), (5, 16, 8, 6, 96, 96), (2, 2, 3, 4, 40, 64), (2, 2, 8, 5, 96, 96),
                                            (5, 16, 1, 2, 7, 40), (5, 16, 40, 3, 64, 64)])
def test_convlstm_layer_sequences(cin, F, B, T, H, W, hip_ops, ref_ops):
    """ConvLSTM2D(return_sequences=True) over T > 1 steps (models.py:93,101): the layer's per-timestep launches on the HIP
    backend (fused recurrent steps, T-batched input convolution, time loops replayed from HIP graphs) against the same
    layer program on the oracle backend: h, the input gradient, and the kernel / recurrent kernel / bias gradients."""
    from downscaling.engine.layers import ConvLSTM
    from downscaling.engine.params import ParamStore
    dev = hip_ops.device
    gen = torch.Generator().manual_seed(41)
    big = B * T * H * W > 100_000
    wscale = 0.02 if big else 0.25                   # big cases: pre-activations stay off the hard-sigmoid knots
    ldx, ldh = (cin + 3) // 4 * 4, (F + 3) // 4 * 4
    x64 = torch.zeros(T * B, H, W, ldx, dtype=torch.float64)
    x64[..., :cin] = torch.randn(T * B, H, W, cin, generator=gen, dtype=torch.float64)
    dh64 = torch.zeros(T * B, H, W, ldh, dtype=torch.float64)
    dh64[..., :F] = torch.randn(T * B, H, W, F, generator=gen, dtype=torch.float64)
    out = {}
    for tag, ops, cv in (("ref", ref_ops, lambda t: t.clone()), ("hip", hip_ops, lambda t: t.float().to(dev).contiguous())):
        class Net:
            pass
        net = Net()
        net.ops, net.params = ops, ParamStore(ops)
        layer = ConvLSTM(net, "l", cin, F)
        net.params.finalize(np.random.default_rng(5))
        rng = np.random.default_rng(6)
        net.params.set_weights({v.name: rng.normal(0, wscale, v.shape) for v in net.params.vars})
        layer.build()
        h = ops.zeros(T * B, H, W, ldh)
        x, dh = cv(x64), cv(dh64)
        layer.forward(x, h, B, T)
        dx = ops.zeros(T * B, H, W, ldx)
        net.params.zero_grad()
        layer.backward(x, h, dh, dx, B, T, need_wgrad=True)
        out[tag] = dict(h=h.clone(), dx=dx, gx=layer.wx.grad.clone(), gh=layer.wh.grad.clone(), gb=layer.b.grad.clone())
    assert rel_err(out["hip"]["h"], out["ref"]["h"]) < 5 * TOL
    assert float(out["hip"]["h"][..., F:].abs().max() if ldh > F else 0.0) == 0.0
    # The hard sigmoid's derivative jumps at its two knots: a pre-activation that sits within fp32 rounding of a knot takes
    # the other branch than the fp64 oracle (a measure-zero event per value, ~1 value per 10^7), and that single dgates
    # entry then spreads through the recurrence.  So the gradients are compared by quantile, with the maximum bounded.
    for k in ("dx", "gx", "gh", "gb"):
        a, b = out["hip"][k].double().cpu().flatten(), out["ref"][k].flatten()
        err = (a - b).abs() / b.abs().max()
        q = float(torch.quantile(err[:: max(1, err.numel() // 4_000_000)], 0.999)) if err.numel() > 10 else float(err.max())
        assert q < 5 * TOL, (k, q)
        assert float(err.max()) < 0.2, (k, float(err.max()))


@pytest.mark.parametrize("F,cinp,n,H,W", [(16, 16, 3, 24, 40), (16, 16, 8, 96, 96), (2, 4, 3, 24, 40), (2, 4, 8, 96, 96), (16, 16, 2, 19, 33),
                                          (128, 128, 8, 24, 24), (128, 128, 2, 17, 13), (32, 32, 3, 8, 24)])
def test_convlstm_recurrent_step_fused(hip_ops, ref_ops, F, cinp, n, H, W):
    """fp32 ConvLSTM recurrent step in one launch (wdg_convlstm_step: halo-tile convolution with the cell update in its
    epilogue; the discriminator's 16- and 2-feature ConvLSTMs at n_timesteps > 1, models.py:93,101 — and
    wdg_convlstm_step_gemm: the generator's 128-feature layer, models.py:45, through the implicit GEMM with gate-interleaved
    weight rows) against the oracle's accumulating convolution + cell update: pre-activation gates (read by the backward
    pass), c_t and h_t."""
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    g, rg = ConvGeom(3, 3, 1, 1), RG(3, 3, 1, 1)
    gen = torch.Generator().manual_seed(F + H)
    dev = hip_ops.device
    h_prev = torch.zeros(n, H, W, cinp, dtype=torch.float64)
    h_prev[..., :F] = torch.randn(n, H, W, F, generator=gen, dtype=torch.float64)
    gates = torch.randn(n, H, W, 4 * F, generator=gen, dtype=torch.float64) * 2
    c_prev = torch.randn(n, H, W, F, generator=gen, dtype=torch.float64)
    w = torch.randn(3, 3, F, 4 * F, generator=gen, dtype=torch.float64) * 0.3
    pk_r, pk_g = ref_ops.pack_weights(w), hip_ops.pack_weights(w.float().to(dev).contiguous())
    g_r, c_r, h_r = gates.clone(), torch.zeros_like(c_prev), torch.zeros_like(h_prev)
    ref_ops.conv_fwd(h_prev, pk_r, None, g_r, rg, act=False, accumulate=True)
    ref_ops.lstm_fwd(g_r.view(-1, 4 * F), c_prev.view(-1, F), c_r.view(-1, F), h_r.view(-1, cinp), F)
    g_g, hp_g, cp_g = gates.float().to(dev), h_prev.float().to(dev), c_prev.float().to(dev)
    c_g, h_g = torch.zeros(n, H, W, F, device=dev), torch.zeros(n, H, W, cinp, device=dev)
    assert hip_ops.convlstm_step_supported(hp_g, g_g, pk_g, g, F)
    # 16 features: the step kernels of csrc/convlstm16.hip (default) and the halo-tile kernel's cell epilogue behind them
    for own16 in ((1, 0) if F == 16 else (1,)):
        assert hip_ops.lib.wdg_set_tuning(b"lstm16_step", own16) == 0
        try:
            g_g, c_g, h_g = gates.float().to(dev), torch.zeros(n, H, W, F, device=dev), torch.zeros(n, H, W, cinp, device=dev)
            hip_ops.convlstm_step(hp_g, pk_g, g_g, cp_g, c_g, h_g, g, F)
        finally:
            hip_ops.lib.wdg_set_tuning(b"lstm16_step", 1)
        assert rel_err(g_g, g_r) < TOL, "pre-activation gates"
        # a hard-sigmoid knot or tanh amplifies nothing here: c and h are smooth in the gates up to the clip points
        assert rel_err(c_g, c_r) < 5 * TOL and rel_err(h_g[..., :F], h_r[..., :F]) < 5 * TOL
        assert float(h_g[..., F:].abs().max()) == 0.0 if cinp > F else True


@pytest.mark.parametrize("n,H,W", [(3, 19, 33), (5, 96, 96), (2, 8, 32), (1, 41, 70)])
def test_convlstm_gates_x(hip_ops, ref_ops, n, H, W):
    """Input part of the 5 -> 16-feature ConvLSTM's gate pre-activations for all timesteps (wdg_convlstm_gates_x: the matrix-pipe
    kernel of the single-timestep layer storing its four accumulators; models.py:101 at n_timesteps > 1) against the oracle's
    convolution + bias, ragged maps included."""
    from oracle.torch_backend import ConvGeom as RG
    gen = torch.Generator().manual_seed(H * W)
    dev = hip_ops.device
    x = torch.zeros(n, H, W, 8, dtype=torch.float64)
    x[..., :5] = torch.randn(n, H, W, 5, generator=gen, dtype=torch.float64)
    w = torch.randn(3, 3, 5, 64, generator=gen, dtype=torch.float64) * 0.3
    b = torch.randn(64, generator=gen, dtype=torch.float64)
    y_r = torch.zeros(n, H, W, 64, dtype=torch.float64)
    ref_ops.conv_fwd(x, ref_ops.pack_weights(w), b, y_r, RG(3, 3, 1, 1), act=False)
    y_g = torch.full((n, H, W, 64), float("nan"), device=dev)
    x_g = x.float().to(dev)
    x_g[..., 5:] = 7.0          # the pad channels of the stored input are not part of the layer
    assert hip_ops.convlstm_gates_x_supported(x_g, y_g, 5, 16)
    hip_ops.convlstm_gates_x(x_g, w.float().to(dev).contiguous(), b.float().to(dev), y_g, 5, 16)
    assert rel_err(y_g, y_r) < TOL


@pytest.mark.parametrize("n,H,W", [(3, 19, 33), (5, 96, 96), (2, 8, 32), (1, 41, 70)])
def test_convlstm_gates_x_two_features(hip_ops, ref_ops, n, H, W):
    """The 2 -> 2-feature layer (models.py:93) at n_timesteps > 1: input part of the gates for all timesteps
    (wdg_convlstm_gates_x, one pixel per thread) and its data gradient (wdg_convlstm_gates_dx: overwrite and accumulate, dx a
    channel-padded view whose pad channels stay untouched) against the oracle's convolution / transposed convolution."""
    from oracle.torch_backend import ConvGeom as RG
    gen = torch.Generator().manual_seed(H * W + 1)
    dev = hip_ops.device
    x = torch.zeros(n, H, W, 4, dtype=torch.float64)
    x[..., :2] = torch.randn(n, H, W, 2, generator=gen, dtype=torch.float64)
    w = torch.randn(3, 3, 2, 8, generator=gen, dtype=torch.float64) * 0.3
    b = torch.randn(8, generator=gen, dtype=torch.float64)
    pk_r = ref_ops.pack_weights(w)
    y_r = torch.zeros(n, H, W, 8, dtype=torch.float64)
    ref_ops.conv_fwd(x, pk_r, b, y_r, RG(3, 3, 1, 1), act=False)
    y_g = torch.full((n, H, W, 8), float("nan"), device=dev)
    x_g = x.float().to(dev)
    x_g[..., 2:] = 7.0          # the pad channels of the stored input are not part of the layer
    w_g = w.float().to(dev).contiguous()
    assert hip_ops.convlstm_gates_x_supported(x_g, y_g, 2, 2)
    hip_ops.convlstm_gates_x(x_g, w_g, b.float().to(dev), y_g, 2, 2)
    assert rel_err(y_g, y_r) < TOL
    dg = torch.randn(n, H, W, 8, generator=gen, dtype=torch.float64)
    dx_r = torch.zeros(n, H, W, 4, dtype=torch.float64)
    ref_ops.conv_dgrad(dg, pk_r, dx_r, RG(3, 3, 1, 1))
    dg_g = dg.float().to(dev)
    dx_g = torch.full((n, H, W, 4), 5.0, device=dev)
    assert hip_ops.convlstm_gates_dx_supported(dg_g, dx_g, 2, 2)
    hip_ops.convlstm_gates_dx(dg_g, w_g, dx_g, 2, 2, accumulate=False)
    assert rel_err(dx_g[..., :2], dx_r[..., :2]) < TOL and float((dx_g[..., 2:] - 5.0).abs().max()) == 0.0
    hip_ops.convlstm_gates_dx(dg_g, w_g, dx_g, 2, 2, accumulate=True)
    assert rel_err(dx_g[..., :2], 2 * dx_r[..., :2]) < TOL


@pytest.mark.parametrize("F,cinp,n,H,W,first", [(16, 16, 3, 24, 40, False), (16, 16, 8, 96, 96, True), (2, 4, 3, 24, 40, False),
                                                (2, 4, 8, 96, 96, True), (16, 16, 2, 19, 33, False)])
def test_convlstm_recurrent_bwd_step_fused(hip_ops, ref_ops, F, cinp, n, H, W, first):
    """Backward counterpart (wdg_convlstm_bwd_step): dh_{t-1} += conv_transpose(dgates_t) and the cell backward of timestep t-1
    in the same launch, against the oracle's data gradient + wdg_lstm_bwd restatement.  first=True: t-1 = 0 (no c_prev, no
    dc flowing on)."""
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    g, rg = ConvGeom(3, 3, 1, 1), RG(3, 3, 1, 1)
    gen = torch.Generator().manual_seed(7 * F + W)
    dev = hip_ops.device
    rn = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64)
    dg_next, gates = rn(n, H, W, 4 * F), rn(n, H, W, 4 * F) * 2
    dh = torch.zeros(n, H, W, cinp, dtype=torch.float64)
    dh[..., :F] = rn(n, H, W, F)
    c_prev, c_cur, dc_in = (None if first else rn(n, H, W, F)), rn(n, H, W, F), rn(n, H, W, F)
    w = rn(3, 3, F, 4 * F) * 0.3
    pk_r, pk_g = ref_ops.pack_weights(w), hip_ops.pack_weights(w.float().to(dev).contiguous())
    dh_r, dg_r = dh.clone(), torch.zeros(n, H, W, 4 * F, dtype=torch.float64)
    dc_r = None if first else torch.zeros(n, H, W, F, dtype=torch.float64)
    ref_ops.conv_dgrad(dg_next, pk_r, dh_r, rg, accumulate=True)
    ref_ops.lstm_bwd(gates.view(-1, 4 * F), None if first else c_prev.view(-1, F), c_cur.view(-1, F), dh_r.view(-1, cinp),
                     dc_in.view(-1, F), dg_r.view(-1, 4 * F), None if first else dc_r.view(-1, F), F)
    f32 = lambda t: None if t is None else t.float().to(dev)
    dh_g, dg_g = f32(dh), torch.zeros(n, H, W, 4 * F, device=dev)
    dc_g = None if first else torch.zeros(n, H, W, F, device=dev)
    assert hip_ops.convlstm_bwd_step_supported(dh_g, dg_g, pk_g, g, F)
    for own16 in ((1, 0) if F == 16 else (1,)):          # (csrc/convlstm16.hip, then the halo-tile kernel's epilogue)
        assert hip_ops.lib.wdg_set_tuning(b"lstm16_step", own16) == 0
        try:
            dh_g, dg_g = f32(dh), torch.zeros(n, H, W, 4 * F, device=dev)
            dc_g = None if first else torch.zeros(n, H, W, F, device=dev)
            hip_ops.convlstm_bwd_step(f32(dg_next), pk_g, dh_g, f32(gates), f32(c_prev), f32(c_cur), f32(dc_in), dg_g, dc_g, g, F)
        finally:
            hip_ops.lib.wdg_set_tuning(b"lstm16_step", 1)
        assert rel_err(dh_g, dh_r) < TOL, "dh"
        # the hard-sigmoid gradient is a step function: fp32 vs fp64 pre-activations may sit on different sides of a knot for a few
        # elements, so dgates is compared robustly (99.9 % quantile) and in the mean
        d = (dg_g.double().cpu() - dg_r).abs().flatten()
        scale = float(dg_r.abs().max())
        assert float(torch.quantile(d[:2_000_000], 0.999)) < 20 * TOL * scale and float(d.mean()) < TOL * scale, "dgates"
        if not first:
            assert rel_err(dc_g, dc_r) < 10 * TOL, "dc"


@pytest.mark.parametrize("P,C,ld", [(100003, 64, 64), (5000, 4, 4), (7777, 8, 8), (9001, 16, 16), (30011, 128, 160), (4097, 512, 512),
                                    (2500, 12, 12), (3001, 2, 4), (70000, 64, 68), (5, 64, 64)])
def test_colsum_shapes(hip_ops, ref_ops, P, C, ld):
    """Bias gradients (column sums over pixels): few and many channels, dense and strided rows, ragged row counts, overwrite and
    accumulate.  (A 16-byte-per-lane form of the kernel was measured no faster — T = 24 step 107.1 vs 106.2 ms — and dropped.)"""
    gen = torch.Generator().manual_seed(P + C)
    x = torch.randn(P, ld, generator=gen, dtype=torch.float64)
    xg = x.float().to(hip_ops.device)
    for accumulate in (False, True):
        o_r = torch.linspace(-1, 1, C, dtype=torch.float64)
        o_g = o_r.float().to(hip_ops.device)
        ref_ops.colsum(x[:, :C].float().double(), o_r, accumulate=accumulate)
        hip_ops.colsum(xg[:, :C], o_g, accumulate=accumulate)
        assert float((o_g.double().cpu() - o_r).abs().max()) < 2e-5 * max(1.0, float(o_r.abs().max())) * max(1.0, (P / 1e4) ** 0.5), accumulate
