"""The opt-in bf16-slice mode of the implicit-GEMM kernels (conv_igemm.hip PIPE 4, HipOps.set_split_mode): every fp32
operand is the exact sum of three bf16 slices, six slice products on the bf16 MFMA stand for the fp32 product.  The mode
claims fp32-rounding accuracy, so it has to pass the SAME bound as the fp32-MFMA kernels against the fp64 restatement
(2e-5, tests/test_ops_gpu.py), stay within a few fp32 ulps of the fp32-MFMA result, and follow weight updates (the
pre-sliced weight copies are refreshed when the packed weights change).  The whole GPU suite also passes with WDG_SPLIT=1
(profiles/r02v_gpu_tests_split.log); this file keeps the mode inside the default `-m gpu` run."""
import pytest
import torch

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-5

CASES = [
    # name, n, H, W, cin, cout, k, s, p      (tile shapes: 128x128 / 128x64 with pre-sliced weights, 64x64 and 256x32 without)
    ("g0_8x8s2_cin23", 2, 64, 64, 23, 128, 8, 2, 3),
    ("gates_128to512", 2, 32, 32, 128, 512, 3, 1, 1),
    ("d_7x7s3", 3, 40, 40, 32, 64, 7, 3, 1),
    ("d_7x7s3_odd_small", 2, 27, 27, 64, 128, 7, 3, 1),
    ("convT2x2_as_conv", 2, 32, 32, 32, 192, 2, 2, 0),
    ("col_gemm_1x1", 2, 32, 32, 400, 160, 1, 1, 0),
    ("cin_not_mult8", 2, 24, 24, 12, 72, 3, 1, 1),
]


@pytest.fixture()
def split_ops(hip_ops):
    before = hip_ops.split_mode             # (True when the whole suite runs with WDG_SPLIT=1)
    hip_ops.set_split_mode(True)
    try:
        yield hip_ops
    finally:
        hip_ops.set_split_mode(before)


def _mk(case, seed=3):
    name, n, H, W, cin, cout, k, s, p = case
    gen = torch.Generator().manual_seed(seed)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    cin_p, cout_p = (cin + 3) // 4 * 4, (cout + 3) // 4 * 4
    x = torch.zeros(n, H, W, cin_p, dtype=torch.float64)
    x[..., :cin] = torch.randn(n, H, W, cin, generator=gen, dtype=torch.float64) * torch.logspace(-3, 3, cin, dtype=torch.float64)
    dy = torch.zeros(n, Ho, Wo, cout_p, dtype=torch.float64)
    dy[..., :cout] = torch.randn(n, Ho, Wo, cout, generator=gen, dtype=torch.float64)
    w = torch.randn(k, k, cin, cout, generator=gen, dtype=torch.float64) * 0.05
    b = torch.randn(cout, generator=gen, dtype=torch.float64)
    return dict(n=n, H=H, W=W, Ho=Ho, Wo=Wo, cin=cin, cout=cout, cin_p=cin_p, cout_p=cout_p, k=k, s=s, p=p, x=x.float().double(),
                dy=dy.float().double(), w=w.float().double(), b=b.float().double())


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_split_mode_conv_parity(case, split_ops, ref_ops):
    """Forward and data gradient in slice mode vs the fp64 restatement (same bound as the fp32-MFMA kernels) and vs the
    fp32-MFMA kernels themselves; inputs span six decades per channel so that a lost slice would show."""
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    ops = split_ops
    d = _mk(case)
    g, rg = ConvGeom(d["k"], d["k"], d["s"], d["p"]), RG(d["k"], d["k"], d["s"], d["p"])
    dev = ops.device
    pk_g, pk_r = ops.pack_weights(d["w"].float().to(dev).contiguous()), ref_ops.pack_weights(d["w"])
    x_g, dy_g, b_g = d["x"].float().to(dev), d["dy"].float().to(dev), d["b"].float().to(dev)
    y_r = torch.zeros(d["n"], d["Ho"], d["Wo"], d["cout_p"], dtype=torch.float64)
    ref_ops.conv_fwd(d["x"], pk_r, d["b"], y_r, rg, act=True)
    y_s = torch.zeros_like(y_r, dtype=torch.float32, device=dev)
    ops.conv_fwd(x_g, pk_g, b_g, y_s, g, act=True)
    assert rel_err(y_s, y_r) < TOL, "fwd (slice mode) vs fp64"
    dx_r = torch.zeros(d["n"], d["H"], d["W"], d["cin_p"], dtype=torch.float64)
    ref_ops.conv_dgrad(d["dy"], pk_r, dx_r, rg)
    dx_s = torch.zeros_like(dx_r, dtype=torch.float32, device=dev)
    ops.conv_dgrad(dy_g, pk_g, dx_s, g)
    assert rel_err(dx_s, dx_r) < TOL, "dgrad (slice mode) vs fp64"
    ops.set_split_mode(False)
    y_f, dx_f = torch.zeros_like(y_s), torch.zeros_like(dx_s)
    ops.conv_fwd(x_g, pk_g, b_g, y_f, g, act=True)
    ops.conv_dgrad(dy_g, pk_g, dx_f, g)
    ops.set_split_mode(True)
    assert rel_err(y_s, y_f) < 2e-6 and rel_err(dx_s, dx_f) < 2e-6, "slice mode vs fp32 MFMA"
    # the slice-mode error is of the size of the fp32-MFMA kernel's own (both are fp32 accumulations of exact-ish products)
    assert rel_err(y_s, y_r) < 4 * max(rel_err(y_f, y_r), 1e-7)


def test_split_mode_follows_weight_updates(split_ops, ref_ops):
    """The pre-sliced weight copy is refreshed when the packed weights change (PackedWeights.mark_stale via refresh / the
    prep batch), and a new PackedWeights at a recycled address never sees another object's copy."""
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    ops, dev = split_ops, split_ops.device
    g, rg = ConvGeom(3, 3, 1, 1), RG(3, 3, 1, 1)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(2, 32, 32, 128, generator=gen, dtype=torch.float64).float().double()
    x_g = x.float().to(dev)
    for rep in range(3):
        w = (torch.randn(3, 3, 128, 128, generator=gen, dtype=torch.float64) * 0.05).float().double()
        w_g = w.float().to(dev).contiguous()
        pk = ops.pack_weights(w_g)
        for upd in range(2):
            y_r = torch.zeros(2, 32, 32, 128, dtype=torch.float64)
            ref_ops.conv_fwd(x, ref_ops.pack_weights(w), None, y_r, rg)
            y_g = torch.zeros(2, 32, 32, 128, device=dev)
            ops.conv_fwd(x_g, pk, None, y_g, g)
            assert rel_err(y_g, y_r) < TOL, (rep, upd)
            w = (w * 0.5 + 0.01).float().double()          # in-place update of the master weights, then repack
            w_g.copy_(w.float().to(dev))
            pk.refresh()
        del pk, w_g
