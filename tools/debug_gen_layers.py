#!/usr/bin/env python3
"""Layer-by-layer comparison of GeneratorNet on the HIP backend (fp32) against the same program on the oracle backend (fp64):
    python tools/debug_gen_layers.py S B F T [training]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402


def main():
    S, B, F, T = (int(a) for a in sys.argv[1:5])
    training = len(sys.argv) <= 5 or sys.argv[5] != "0"
    from downscaling.engine.hipops import HipOps
    from downscaling.engine.networks import GeneratorNet
    from oracle.torch_backend import TorchOps
    from tests.helpers import randomize
    hip, ref = HipOps("cuda:0"), TorchOps(torch.float64)
    nets = [GeneratorNet(o, S, 3, 4, 2, T, feature_channels=F, seed=3) for o in (hip, ref)]
    for n in nets:
        randomize(n, 11)
    g = torch.Generator().manual_seed(0)
    low = torch.randn(B, T, S, S, 3, generator=g, dtype=torch.float64)
    noise = torch.randn(B, T, S, S, 4, generator=g, dtype=torch.float64) * 0.1
    for n, o in zip(nets, (hip, ref)):
        n.set_image(low.to(o.device, o.dtype))
        n.set_noise(noise.to(o.device, o.dtype))
        n.forward(B, training)
    bh, br = nets[0].buffers(B), nets[1].buffers(B)
    for k in ("x0", "y0", "cat2", "y2", "cat4", "h", "y5", "y7", "y9", "z9", "out"):
        a, r = bh[k].double().cpu(), br[k].double()
        err = float((a - r).abs().max() / max(1e-30, float(r.abs().max())))
        print(f"{k:6s} shape {tuple(r.shape)} rel err {err:.3e}   max |ref| {float(r.abs().max()):.3e}")
    if training:
        gout = torch.randn(B, T, S, S, 2, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
        for n, o in zip(nets, (hip, ref)):
            dout = o.zeros(T * B, S, S, 4)
            n.to_time_major(gout.to(o.device, o.dtype), dout)
            n.params.zero_grad()
            n.backward(B, dout)
        for k in ("x0", "y0", "cat2", "y2", "cat4", "h", "y5", "y7", "y9", "z9", "out"):
            a, r = bh[k].double().cpu(), br[k].double()
            nimg = r.shape[0]
            peri = [float((a[i] - r[i]).abs().max() / max(1e-30, float(r.abs().max()))) for i in range(nimg)]
            print(f"after backward {k:6s} rel err {max(peri):.3e}  worst image {peri.index(max(peri))}")
        gh, gr = nets[0].grad_buffers(B), nets[1].grad_buffers(B)
        for k in gh:
            a, r = gh[k].double().cpu(), gr[k].double()
            print(f"grad buffer {k:6s} rel err {float((a - r).abs().max() / max(1e-30, float(r.abs().max()))):.3e}")
            C = r.shape[-1]
            per = [float((a[..., c] - r[..., c]).abs().max() / max(1e-30, float(r.abs().max()))) for c in range(C)]
            print("      per channel: " + " ".join(f"{e:.0e}" for e in per))
            nimg = r.shape[0]
            peri = [float((a[i] - r[i]).abs().max() / max(1e-30, float(r.abs().max()))) for i in range(nimg)]
            print("      per image (first 16, last 4): " + " ".join(f"{e:.0e}" for e in peri[:16] + peri[-4:]))
        if getattr(nets[0].lstm, "dgates", None) is not None:
            a, r = nets[0].lstm.dgates.double().cpu(), nets[1].lstm.dgates.double()
            print("lstm dgates rel err %.3e" % float((a - r).abs().max() / r.abs().max()))
        for va, vr in zip(nets[0].params.vars, nets[1].params.vars):
            if va.grad is None or vr.grad is None:
                continue
            a, r = va.grad.double().cpu(), vr.grad.double()
            err = float((a - r).abs().max() / max(1e-30, float(r.abs().max())))
            # the same comparison against a noise model: fp32 sums of the per-pixel products -> eps * sqrt(terms) * |terms|
            print(f"dW {va.name:50s} rel err {err:.3e}  max|ref| {float(r.abs().max()):.3e}")
    if hasattr(nets[0].lstm, "gates") and nets[0].lstm.gates is not None:
        a, r = nets[0].lstm.gates.double().cpu(), nets[1].lstm.gates.double()
        print("lstm gates rel err %.3e" % float((a - r).abs().max() / r.abs().max()))
        for q, nm in enumerate("ifco"):
            aa, rr = a[..., q * F:(q + 1) * F], r[..., q * F:(q + 1) * F]
            print(f"   gate {nm}: {float((aa - rr).abs().max() / max(1e-30, float(rr.abs().max()))):.3e}")


if __name__ == "__main__":
    main()
