#!/usr/bin/env python3
"""GeneratorNet.backward's first operations replayed step by step on the HIP backend and on the oracle backend (same program,
fp64), comparing the buffer each step writes:  python tools/debug_gen_bwd_steps.py S B F"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402
from downscaling.engine.common import v2  # noqa: E402


def main():
    S, B, F = (int(a) for a in sys.argv[1:4])
    T = 1
    from downscaling.engine.hipops import HipOps
    from downscaling.engine.networks import GeneratorNet
    from oracle.torch_backend import TorchOps
    from tests.helpers import randomize
    hip, ref = HipOps("cuda:0"), TorchOps(torch.float64)
    nets = [GeneratorNet(o, S, 3, 4, 2, T, feature_channels=F, seed=3) for o in (hip, ref)]
    for n in nets:
        randomize(n, 11)
        n.wgrad_stream = False
    g = torch.Generator().manual_seed(0)
    low = torch.randn(B, T, S, S, 3, generator=g, dtype=torch.float64)
    noise = torch.randn(B, T, S, S, 4, generator=g, dtype=torch.float64) * 0.1
    gout = torch.randn(B, T, S, S, 2, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    douts = []
    for n, o in zip(nets, (hip, ref)):
        n.set_image(low.to(o.device, o.dtype))
        n.set_noise(noise.to(o.device, o.dtype))
        n.forward(B, True)
        dout = o.zeros(T * B, S, S, 4)
        n.to_time_major(gout.to(o.device, o.dtype), dout)
        n.params.zero_grad()
        douts.append(dout)

    def cmp(label, fa, fr):
        a, r = fa.double().cpu(), fr.double()
        nimg = r.shape[0]
        peri = [float((a[i] - r[i]).abs().max() / max(1e-30, float(r.abs().max()))) for i in range(nimg)]
        C = r.shape[-1]
        perc = [float((a[..., c] - r[..., c]).abs().max() / max(1e-30, float(r.abs().max()))) for c in range(C)]
        print(f"{label:42s} rel err {max(peri):.3e} worst image {peri.index(max(peri))}; per channel " + " ".join(f"{e:.0e}" for e in perc[:24]))

    def both(fn):
        return [fn(n, n.buffers(B), n.grad_buffers(B), d) for n, d in zip(nets, douts)]

    r = both(lambda n, b, g, d: (n.c11.backward_input(d, g["dz9"]), g["dz9"])[1]); cmp("c11.backward_input -> dz9", *r)
    r = both(lambda n, b, g, d: (n.bn10.backward(v2(g["dz9"]), v2(b["y9"]), v2(g["dz9"]), n.c9.b.grad_pad), g["dz9"])[1]); cmp("bn10.backward (in place)", *r)
    r = both(lambda n, b, g, d: (n.ops.upconv_bwd(b["cat2"], g["dz9"], n.c9.pk, n.c9.w.grad, g["dcat2"], n.c9.g, pool=n._scratch_pool(b)), g["dcat2"])[1]); cmp("upconv_bwd -> dcat2", *r)
    # variants of the failing call on copies of its inputs
    def bn8_variant(n, b, g, d, mode):
        o = n.ops
        dz = v2(g["dcat2"][..., :n.F4p])
        if mode == "contig_out":
            src, dst = dz, o.empty(*dz.shape)
        elif mode == "contig_in_place":
            src = dst = dz.clone()
        elif mode == "wide_copy_in_place":
            wide = g["dcat2"].clone()
            src = dst = v2(wide[..., :n.F4p])
        bn = n.bn8
        red = o.zeros(2 * bn.Cp, dtype=torch.float64)
        o.bn_bwd_reduce(src, v2(b["y7"]), bn.saved, red)
        o.bn_bwd_apply(src, v2(b["y7"]), bn.saved, bn.gamma.value_pad, red, red, bn.count, 0.2, dst, None, None, None)
        return dst.reshape(g["dcat2"].shape[0], g["dcat2"].shape[1], g["dcat2"].shape[2], -1).clone()
    for mode in ("contig_out", "contig_in_place", "wide_copy_in_place"):
        r = both(lambda n, b, g, d: bn8_variant(n, b, g, d, mode)); cmp("bn8 variant " + mode, *r)
    print("bn8: count", nets[0].bn8.count, nets[1].bn8.count, "saved", nets[0].bn8.saved.cpu().tolist(), nets[1].bn8.saved.tolist())
    print("gamma_pad", nets[0].bn8.gamma.value_pad.cpu().tolist(), nets[1].bn8.gamma.value_pad.tolist())
    r = both(lambda n, b, g, d: (n.bn8.backward(v2(g["dcat2"][..., :n.F4p]), v2(b["y7"]), v2(g["dcat2"][..., :n.F4p]), n.c7.b.grad_pad), g["dcat2"])[1]); cmp("bn8.backward (in place on dcat2[:F4p])", *r)
    r = both(lambda n, b, g, d: (n.c7.backward_input(g["dcat2"][..., :n.F4p], g["dcat4"]), g["dcat4"])[1]); cmp("c7.backward_input -> dcat4", *r)
    r = both(lambda n, b, g, d: (n.bn6.backward(v2(g["dcat4"][..., :F // 2]), v2(b["y5"]), v2(g["dcat4"][..., :F // 2]), n.c5.b.grad_pad), g["dcat4"])[1]); cmp("bn6.backward (in place on dcat4[:F/2])", *r)
    r = both(lambda n, b, g, d: (n.c5.backward_input(g["dcat4"][..., :F // 2], g["dh"]), g["dh"])[1]); cmp("c5.backward_input -> dh", *r)


if __name__ == "__main__":
    main()
