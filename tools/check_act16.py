#!/usr/bin/env python3
"""Inference precision: the shipped generator G(96,3,20,2,T=24) on 16 tiles with the 16-bit activation hand-over (WDG_ACT16,
default) against fp32 activations between the same 16-bit layers — the readers round to the operand format either way, so the
outputs must be the same bits.   python tools/check_act16.py [bf16|fp16]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    from downscaling.engine import runtime
    from downscaling.gan.models import make_generator
    ops = runtime.get_ops()
    g = make_generator(96, 3, 20, 2, 24)
    g.graph_inference = False
    torch.manual_seed(1)
    tiles = torch.randn(16, 24, 96, 96, 3, device=ops.device)
    noise = torch.randn(16, 24, 96, 96, 20, device=ops.device) * 0.1
    outs = {}
    for a in (True, False):
        ops.act16 = a
        outs[a] = g([tiles, noise], precision=prec).clone()
    ref = g([tiles, noise], precision="fp32")
    same = torch.equal(outs[True], outs[False])
    err = float((outs[True] - ref).abs().max() / ref.abs().max())
    print(f"{prec}: act16 on/off bit-identical: {same}; max deviation from the fp32 path {err:.2e} of the output range; finite {bool(torch.isfinite(outs[True]).all())}")
    assert same


if __name__ == "__main__":
    main()
