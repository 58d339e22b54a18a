"""Parity at BASELINE.json's FULL sizes (configs[1]: batch 32, 256x256, fp32), where the fp64 oracle is too slow to run
whole: size-independent properties of the domain instead.

  * crops: a convolution output depends only on the receptive field, so a band of rows of the full-size HIP result must
    equal the oracle's result on the corresponding input band (checks the large-offset indexing: image strides, 32-bit
    buffer offsets, tile tails — everything the small-size operator tests cannot reach);
  * adjointness: <conv(x; w), dy> = <x, dgrad(dy; w)> = <w, wgrad(x, dy)> — the three kernels of a layer agree with each
    other at the headline shape, with the output itself as cotangent so that the reference value |y|^2 has no
    cancellation (the forward is pinned by the crop check and by the small-size oracle tests);
  * linearity of the forward in x;
  * the fused upsample + transposed-conv block: forward crop against the oracle, and its column-form backward through
    the same bilinear identities <y, dy> = <x, dx> = <w, dw>;
  * batch consistency of the generator (inference mode): batch 32 equals the same samples in batches of 8;
  * reproducibility: two identical train steps from identical state agree to within the 1e-4 tolerance of the north star (the convolution / SN / split-K sums
    are fixed-order; the BatchNorm statistics and LayerNorm parameter gradients use atomics, so not bit for bit).

Tolerances are those of the operator tests (fp32 rounding: 2e-5 relative for values, 1e-4 for the inner products of
1e8-term sums accumulated in fp32 by different kernels).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
B, S = 32, 256

# name, H, cin, cout, k, stride, pad  (layers of G(256,3,20,2) and D(256,256,3,2) at batch 32)
LAYERS = [
    ("G0 8x8s2 23->128", 256, 23, 128, 8, 2, 3),
    ("G2 4x4s2 128->128", 128, 128, 128, 4, 2, 1),
    ("G4 3x3 128->512 (ConvLSTM gates)", 64, 128, 512, 3, 1, 1),
    ("G5 3x3 128->64", 64, 128, 64, 3, 1, 1),
    ("G11 3x3 16->2", 256, 16, 2, 3, 1, 1),
    ("D conv_b 3x3 16->16", 256, 16, 16, 3, 1, 1),
    ("D 7x7s3 32->64", 256, 32, 64, 7, 3, 1),
    ("D 7x7s3 64->128", 84, 64, 128, 7, 3, 1),
    ("D 7x7s3 256->512", 8, 256, 512, 7, 3, 1),
]


def _dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("name,H,cin,cout,k,s,p", LAYERS, ids=[c[0] for c in LAYERS])
def test_conv_layer_at_headline_shape(name, H, cin, cout, k, s, p, hip_ops, ref_ops):
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    ops, dev = hip_ops, hip_ops.device
    g = ConvGeom(k, k, s, p)
    gen = torch.Generator(device="cpu").manual_seed(1234)
    cp, op = (cin + 3) // 4 * 4, (cout + 3) // 4 * 4
    Ho = (H + 2 * p - k) // s + 1
    x = torch.zeros(B, H, H, cp, device=dev)
    x[..., :cin] = torch.randn(B, H, H, cin, generator=gen).to(dev)
    w = (torch.randn(k, k, cin, cout, generator=gen) * 0.05).to(dev).contiguous()
    pk = ops.pack_weights(w)
    y, dx, dw = ops.zeros(B, Ho, Ho, op), ops.zeros(B, H, H, cp), ops.zeros(k, k, cin, cout)
    ops.conv_fwd(x, pk, None, y, g, act=False)
    dy = y.clone()      # the output itself as cotangent: <y, y> = |y|^2 is large and positive (no cancellation)
    ops.conv_dgrad(dy, pk, dx, g)
    ops.conv_wgrad(x, dy, pk, dw, g, accumulate=False)
    # adjointness of the three kernels
    s1, s2, s3 = _dot(y[..., :cout], dy[..., :cout]), _dot(x[..., :cin], dx[..., :cin]), _dot(w, dw)
    assert abs(s1 - s2) < 1e-4 * s1 and abs(s1 - s3) < 1e-4 * s1, (name, s1, s2, s3)
    # linearity in x
    x2 = torch.zeros_like(x)
    x2[..., :cin] = torch.randn(B, H, H, cin, generator=gen).to(dev)
    y2, y3 = ops.zeros(B, Ho, Ho, op), ops.zeros(B, Ho, Ho, op)
    ops.conv_fwd(x2, pk, None, y2, g, act=False)
    ops.conv_fwd(2.0 * x - 3.0 * x2, pk, None, y3, g, act=False)
    lin = 2.0 * y - 3.0 * y2
    assert float((y3 - lin).abs().max()) < 2e-5 * float(lin.abs().max()), name
    # crop check against the fp64 oracle: output rows [r0, r1) of the LAST image (largest offsets) and of image 0
    r0 = max(0, Ho // 2 - 2)
    r1 = min(Ho, r0 + 4)
    i0, i1 = r0 * s - p, (r1 - 1) * s - p + k           # input rows feeding those output rows
    pad_top, pad_bot = max(0, -i0), max(0, i1 - H)
    for img in (B - 1, 0):
        band = x[img:img + 1, max(i0, 0):min(i1, H), :, :cin].double().cpu()
        band = torch.nn.functional.pad(band, (0, 0, 0, 0, pad_top, pad_bot))
        ref = torch.zeros(1, (band.shape[1] - k) // s + 1, Ho, cout, dtype=torch.float64)
        xin = band.permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(xin, w.double().cpu().permute(3, 2, 0, 1), None, stride=s,
                                         padding=(0, p)).permute(0, 2, 3, 1)
        got = y[img:img + 1, r0:r1, :, :cout].double().cpu()
        assert ref.shape == got.shape, (name, ref.shape, got.shape)
        assert float((got - ref).abs().max()) < 2e-5 * float(ref.abs().max()), (name, img)


def test_upsample_conv_transpose_block_at_headline_shape(hip_ops, ref_ops):
    """models.py:60-64 at batch 32: 128x128x160 -> 256x256x16."""
    from downscaling.engine.hipops import ConvGeom
    from oracle.torch_backend import ConvGeom as RG
    ops, dev = hip_ops, hip_ops.device
    g = ConvGeom(5, 5, 1, 2)
    gen = torch.Generator(device="cpu").manual_seed(77)
    C, N, Hl = 160, 16, 128
    x = torch.randn(B, Hl, Hl, C, generator=gen).to(dev)
    w = (torch.randn(5, 5, N, C, generator=gen) * 0.05).to(dev).contiguous()
    pk = ops.pack_weights(w)
    y = ops.zeros(B, 2 * Hl, 2 * Hl, N)
    ops.upconv_fwd(x, pk, None, y, g, act=False)
    dy = y.clone()      # <y, y> = |y|^2: no cancellation in the reference value
    dw, dx = ops.zeros(5, 5, N, C), ops.zeros(B, Hl, Hl, C)
    ops.upconv_bwd(x, dy, pk, dw, dx, g)
    s1, s2, s3 = _dot(y, dy), _dot(x, dx), _dot(w, dw)
    assert abs(s1 - s2) < 1e-4 * s1 and abs(s1 - s3) < 1e-4 * s1, (s1, s2, s3)
    # forward crop against the oracle: hi-res rows 0..7 (top border classes) and 250..255 of the last image, from
    # the low-res bands that feed them (bilinear 2x + 5x5: 5 hi-res rows of context = 4 low-res rows)
    for lo0, lo1, hr0, hr1 in ((0, 10, 0, 8), (Hl - 10, Hl, 2 * Hl - 6, 2 * Hl)):
        xb = x[B - 1:B, lo0:lo1].double().cpu()
        yr = torch.zeros(1, 2 * (lo1 - lo0), 2 * Hl, N, dtype=torch.float64)
        ref_ops.upconv_fwd(xb, ref_ops.pack_weights(w.double().cpu()), None, yr, RG(5, 5, 1, 2), act=False)
        got = y[B - 1:B, hr0:hr1].double().cpu()
        ref = yr[:, hr0 - 2 * lo0:hr1 - 2 * lo0]
        assert float((got - ref).abs().max()) < 2e-5 * float(ref.abs().max())


def test_generator_batch_consistency_and_train_step_reproducibility(hip_ops):
    """Inference-mode generator: batch 32 equals the same samples in batches of 8 (no cross-sample coupling, all tile /
    split choices differ between the two runs).  Two train steps from identical state agree to 1e-4 relative (observed 0 .. 1e-5)."""
    from downscaling.engine.networks import DiscriminatorNet, GeneratorNet
    from downscaling.engine.trainer import AdamTF, GanEngine, PhiloxSource
    ops, dev = hip_ops, hip_ops.device
    gen = torch.Generator(device="cpu").manual_seed(5)
    low = torch.randn(B, 1, S, S, 3, generator=gen).to(dev)
    noise = (torch.randn(B, 1, S, S, 20, generator=gen) * 0.1).to(dev)
    high = torch.randn(B, 1, S, S, 2, generator=gen).to(dev)
    net = GeneratorNet(ops, S, 3, 20, 2, 1, seed=11)
    net.set_image(low)
    net.set_noise(noise)
    full = net.forward(B, training=False)[..., :2].clone()
    assert bool(torch.isfinite(full).all())
    for i in range(0, B, 8):
        net.set_image(low[i:i + 8])
        net.set_noise(noise[i:i + 8])
        part = net.forward(8, training=False)[..., :2]
        err = float((part - full[i:i + 8]).abs().max()) / float(full.abs().max())
        assert err < 5e-5, (i, err)
    del net
    finals = []
    for _ in range(2):
        g = GeneratorNet(ops, S, 3, 20, 2, 1, seed=11)
        d = DiscriminatorNet(ops, S, S, 3, 2, 1, seed=12)
        eng = GanEngine(g, d, PhiloxSource(ops, seed=99), noise_std=0.1, n_critic=3)
        logs = eng.train_step(low, high, AdamTF(1e-4, 0.5, 0.9, 0.1), AdamTF(4e-4, 0.5, 0.9, 0.1))
        assert all(bool(torch.isfinite(torch.as_tensor(v)).all()) for v in logs.values() if v is not None)
        finals.append((g.params.flat.clone(), d.params.flat.clone(), g.params.state.clone(), d.params.state.clone()))
        del g, d, eng
    for a, b in zip(*finals):
        err = float((a - b).abs().max()) / float(a.abs().max())
        assert err <= 1e-4, err      # observed 0 .. 1e-5 (atomics in the norm statistics / parameter gradients)
