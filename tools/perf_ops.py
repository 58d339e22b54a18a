#!/usr/bin/env python3
"""A/B micro-benchmark of single operator calls at the headline shapes (batch 32, 256 x 256, T = 1) — one process, the
variants interleaved round by round, HIP-event timing of each call on an otherwise idle device.

    python tools/perf_ops.py [--reps 7] [--cases d1_fwd_ln,halo16_dgrad,...] [--variant "name:key=val,key=val"] ...

A variant is a list of wdg_set_tuning(key, val) settings applied before its calls (plans are created per variant where a
knob takes effect at plan creation: --replan).  The first variant is the baseline; the table prints ms per call and
the ratio to the baseline.  Environment switches (WDG_*) are read by HipOps at construction and cannot be A/B'd here.
"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch  # noqa: E402

from downscaling.engine.hipops import ConvGeom, HipOps  # noqa: E402

B = 32


def build_cases(ops, want):
    dev = ops.device
    cases = {}

    def rnd(*shape, scale=1.0):
        return torch.randn(*shape, device=dev) * scale

    def conv_case(tag, n, H, cin, cout, k, s, p, ln=False, ldx=None):
        Ho = (H + 2 * p - k) // s + 1
        cinp, coutp = (cin + 3) // 4 * 4, (cout + 3) // 4 * 4
        xb = rnd(n, H, H, ldx or cinp)
        x = xb[..., :cinp]
        if cin % 4:
            x[..., cin:] = 0
        y, z, dy = rnd(n, Ho, Ho, coutp), rnd(n, Ho, Ho, coutp), rnd(n, Ho, Ho, coutp)
        dx = torch.empty_like(xb)[..., :cinp]
        w = rnd(k, k, cin, cout, scale=0.05).contiguous()
        pk = ops.pack_weights(w)
        dw = torch.zeros_like(w)
        g = ConvGeom(k, k, s, p)
        bias, gamma, beta = rnd(coutp), rnd(coutp), rnd(coutp)
        mr = torch.empty(n * Ho * Ho, 2, device=dev)
        fl = 2.0 * n * Ho * Ho * cout * k * k * cin
        if ln:
            cases[tag + "_fwd_ln"] = (lambda: ops.conv_fwd_ln(x, pk, bias, y, z, g, gamma, beta, 1e-3, mr), fl)
        cases[tag + "_fwd"] = (lambda: ops.conv_fwd(x, pk, bias, y, g, act=True), fl)
        stats = torch.zeros(512, 2 * coutp, dtype=torch.float64, device=dev)
        cases[tag + "_fwd_bn"] = (lambda: ops.conv_fwd(x, pk, bias, y, g, act=True, bn_stats=stats), fl)    # training-mode BatchNorm producer
        cases[tag + "_dgrad"] = (lambda: ops.conv_dgrad(dy, pk, dx, g), fl)
        cases[tag + "_wgrad"] = (lambda: ops.conv_wgrad(x, dy, pk, dw, g, accumulate=True), fl)

    def need(prefix):
        return want is None or any(w.startswith(prefix) for w in want)

    # the discriminator's strided stack (models.py:111-136)
    if need("d1"):
        conv_case("d1", B, 256, 32, 64, 7, 3, 1, ln=True)
    if need("d2"):
        conv_case("d2", B, 84, 64, 128, 7, 3, 1, ln=True)
    if need("d3"):
        conv_case("d3", B, 27, 128, 256, 7, 3, 1, ln=True)
    if need("d4"):
        conv_case("d4", B, 8, 256, 512, 7, 3, 1, ln=True)
    # the generator's layers (models.py:32-71)
    if need("g0"):
        conv_case("g0", B, 256, 23, 128, 8, 2, 3)
    if need("g2"):
        conv_case("g2", B, 128, 128, 128, 4, 2, 1, ldx=160)
    if need("g4"):
        conv_case("g4", B, 64, 128, 512, 3, 1, 1, ldx=192)
    if need("g5"):
        conv_case("g5", B, 64, 128, 64, 3, 1, 1)
    # thin full-resolution layers (halo-tile kernels)
    if need("halo16"):
        conv_case("halo16", B, 256, 16, 16, 3, 1, 1)
        # conv 16 -> 16 + LeakyReLU + LN written into a channel slice of the [hr | mix] concatenation (models.py:102-108)
        x, y, cat = rnd(B, 256, 256, 16), rnd(B, 256, 256, 16), rnd(B, 256, 256, 32)
        w = rnd(3, 3, 16, 16, scale=0.1).contiguous()
        pk = ops.pack_weights(w)
        bias, gamma, beta = rnd(16), rnd(16), rnd(16)
        mr = torch.empty(B * 65536, 2, device=dev)
        g = ConvGeom(3, 3, 1, 1)
        # (default arguments: the names are rebound by the blocks below)
        cases["halo16_fwd_lncat"] = (lambda x=x, pk=pk, bias=bias, y=y, cat=cat, g=g, gamma=gamma, beta=beta, mr=mr:
                                     ops.conv_fwd_ln(x, pk, bias, y, cat[..., 16:], g, gamma, beta, 1e-3, mr), 2.0 * B * 65536 * 16 * 16 * 9)
    if need("g11"):
        conv_case("g11", B, 256, 16, 2, 3, 1, 1)
    if need("lstm"):
        # D's single-timestep ConvLSTMs (models.py:93,101): fused kernels of csrc/convlstm1.hip
        for tag, cin, F in (("lstm_b", 5, 16), ("lstm_a", 2, 2)):
            cinp, Fp = (cin + 3) // 4 * 4, (F + 3) // 4 * 4
            x, h, dh, dx = rnd(B, 256, 256, cinp), torch.zeros(B, 256, 256, Fp, device=dev), rnd(B, 256, 256, Fp), torch.zeros(B, 256, 256, cinp, device=dev)
            x[..., cin:] = 0
            wx, bias = rnd(3, 3, cin, 4 * F, scale=0.2).contiguous(), rnd(4 * F, scale=0.1)
            dw, db = torch.zeros_like(wx), torch.zeros_like(bias)
            fl = 2.0 * B * 65536 * 9 * cin * 3 * F
            cases[tag + "_fwd"] = (lambda x=x, wx=wx, bias=bias, h=h, cin=cin, F=F: ops.convlstm1_fwd(x, wx, bias, h, cin, F), fl)
            cases[tag + "_bwd_x"] = (lambda x=x, wx=wx, bias=bias, dh=dh, dx=dx, cin=cin, F=F: ops.convlstm1_bwd(x, wx, bias, dh, None, dx, cin, F), 2 * fl)
            cases[tag + "_bwd_xw"] = (lambda x=x, wx=wx, bias=bias, dh=dh, dx=dx, cin=cin, F=F, dw=dw, db=db:
                                      ops.convlstm1_bwd(x, wx, bias, dh, None, dx, cin, F, dw=dw, dbias=db), 3 * fl)
            cases[tag + "_bwd_w"] = (lambda x=x, wx=wx, bias=bias, dh=dh, cin=cin, F=F, dw=dw, db=db:
                                     ops.convlstm1_bwd(x, wx, bias, dh, None, None, cin, F, dw=dw, dbias=db), 2 * fl)
    if need("convln"):
        x, z, dz, dx = rnd(B, 256, 256, 4), rnd(B, 256, 256, 32), rnd(B, 256, 256, 32), torch.zeros(B, 256, 256, 4, device=dev)
        x[..., 2:] = 0
        w, bias, gamma, beta = rnd(3, 3, 2, 16, scale=0.3).contiguous(), rnd(16), rnd(16), rnd(16)
        dgm, dbt, dbi, dw = torch.zeros(16, device=dev), torch.zeros(16, device=dev), torch.zeros(16, device=dev), torch.zeros_like(w)
        fl = 2.0 * B * 65536 * 9 * 2 * 16
        cases["convln_fwd"] = (lambda x=x, w=w, bias=bias, gamma=gamma, beta=beta, z=z: ops.convln_fwd(x, w, bias, gamma, beta, 1e-3, 0.2, None, z[..., :16], None), fl)
        cases["convln_bwd_x"] = (lambda x=x, w=w, bias=bias, gamma=gamma, dz=dz, dx=dx: ops.convln_bwd_x(dz[..., :16], x, w, bias, gamma, 1e-3, 0.2, dx, None, None, None, None), 2 * fl)
        cases["convln_bwd_xw"] = (lambda x=x, w=w, bias=bias, gamma=gamma, dz=dz, dx=dx, dgm=dgm, dbt=dbt, dbi=dbi, dw=dw:
                                  ops.convln_bwd_x(dz[..., :16], x, w, bias, gamma, 1e-3, 0.2, dx, dgm, dbt, dbi, dw), 3 * fl)
    if need("upconv"):
        xl = rnd(B, 128, 128, 160)
        w = rnd(5, 5, 16, 160, scale=0.05).contiguous()
        pk = ops.pack_weights(w)
        yo, dy, dxl = torch.empty(B, 256, 256, 16, device=dev), rnd(B, 256, 256, 16), torch.empty(B, 128, 128, 160, device=dev)
        dw = torch.zeros_like(w)
        g = ConvGeom(5, 5, 1, 2)
        bias = rnd(16)
        pool = {}
        fl = 2.0 * B * 16384 * 160 * 400
        cases["upconv_fwd"] = (lambda xl=xl, pk=pk, bias=bias, yo=yo, g=g, pool=pool: ops.upconv_fwd(xl, pk, bias, yo, g, act=True, pool=pool), fl)
        cases["upconv_bwd"] = (lambda xl=xl, dy=dy, pk=pk, dw=dw, dxl=dxl, g=g, pool=pool: ops.upconv_bwd(xl, dy, pk, dw, dxl, g, pool=pool), 2 * fl)
        z = torch.empty(B, 128, 128, 400, device=dev)
        cases["upconv_gemm"] = (lambda xl=xl, pk=pk, z=z: ops.conv_dgrad(xl, pk.as_1x1(), z, ConvGeom(1, 1, 1, 0)), fl)
    if need("ln_bwd"):
        for tag, P, C in (("ln_bwd_c16", B * 65536, 16), ("ln_bwd_c64", B * 84 * 84, 64)):
            dz, y, mr, gm = rnd(P, C), rnd(P, C), torch.rand(P, 2, device=dev) + 0.5, rnd(C)
            dg, db_, dbi = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            cases[tag] = (lambda dz=dz, y=y, mr=mr, gm=gm, dg=dg, db_=db_, dbi=dbi: ops.ln_bwd(dz, y, mr, gm, 0.2, dz, dg, db_, dbi), 0.0)
    if need("bn_bwd"):
        # the generator's BatchNorm backward passes (models.py:34,69): reduce (sum dz, sum dz xh) + apply, on the 16 @ 256^2 and 128 @ 128^2 tensors
        for tag, P, C in (("bn_bwd_c16", B * 65536, 16), ("bn_bwd_c128", B * 16384, 128)):
            dz, y = rnd(P, C), rnd(P, C)
            saved, gm = torch.cat([rnd(C) * 0.1, torch.rand(C, device=dev) + 0.5]), rnd(C)
            red = torch.zeros(2 * C, dtype=torch.float64, device=dev)
            dg, db_, dbi = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            cases[tag + "_reduce"] = (lambda dz=dz, y=y, saved=saved, red=red: ops.bn_bwd_reduce(dz, y, saved, red), 0.0)
            cases[tag + "_apply"] = (lambda dz=dz, y=y, saved=saved, gm=gm, red=red, P=P, dg=dg, db_=db_, dbi=dbi:
                                     ops.bn_bwd_apply(dz, y, saved, gm, red, red, float(P), 0.2, dz, dg, db_, dbi), 0.0)
    if need("misc"):
        flat = rnd(8_600_000)
        offs = torch.tensor([0, 6_422_528, 6_422_528, 8_000_000, 8_000_000, 8_600_000], dtype=torch.int64, device=dev)
        out3 = torch.zeros(3, device=dev)
        cases["misc_meansq"] = (lambda: ops.segment_meansq(flat, offs, out3), 0.0)
        xg, og = rnd(B * 4096, 512), torch.zeros(512, device=dev)
        cases["misc_colsum512"] = (lambda: ops.colsum(xg, og, accumulate=True), 0.0)
        x2c, o2 = rnd(B * 65536, 4), torch.zeros(2, device=dev)
        cases["misc_colsum2"] = (lambda: ops.colsum(x2c[:, :2], o2, accumulate=True), 0.0)
    if want is not None:
        missing = [w for w in want if w not in cases]
        if missing:
            raise SystemExit(f"unknown cases {missing}; known: {sorted(cases)}")
        cases = {k: cases[k] for k in want}
    return cases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--cases", default=None, help="comma-separated case names (default: all)")
    ap.add_argument("--variant", action="append", default=[], help='"name:key=val,key=val" (repeatable; first = baseline; "base:" = no knobs)')
    ap.add_argument("--replan", action="store_true", help="drop the conv plans when switching variants (knobs read at plan creation)")
    args = ap.parse_args()
    ops = HipOps("cuda:0")
    variants = []
    for v in args.variant or ["base:"]:
        name, _, kv = v.partition(":")
        variants.append((name, [(k.encode(), int(x)) for k, x in (it.split("=") for it in kv.split(",") if it)]))
    want = args.cases.split(",") if args.cases else None
    cases = build_cases(ops, want)
    # knobs that are not mentioned by a variant keep the value the previous variant left: every variant should set every knob
    times = {(c, v[0]): [] for c in cases for v in variants}
    for r in range(args.reps + 2):
        for vname, knobs in variants:
            for k, x in knobs:
                if ops.lib.wdg_set_tuning(k, x) != 0:
                    raise SystemExit(f"wdg_set_tuning({k}, {x}) rejected")
            if args.replan:
                ops._plans.clear()
                for cname, (fn, _) in cases.items():      # (plan creation — host work and table uploads — outside the timed calls)
                    fn()
                torch.cuda.synchronize()
            for cname, (fn, _) in cases.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    times[(cname, vname)].append(e0.elapsed_time(e1))
    med = lambda xs: sorted(xs)[len(xs) // 2]     # noqa: E731
    print(f"{'case':20s} " + " ".join(f"{v[0]:>22s}" for v in variants))
    for cname, (_, fl) in cases.items():
        base = med(times[(cname, variants[0][0])])
        row = f"{cname:20s} "
        for vname, _ in variants:
            t = med(times[(cname, vname)])
            row += f" {1e3 * t:8.1f}us {fl / max(t, 1e-9) * 1e-9:6.1f}TF {t / base:5.3f}"
        print(row)


if __name__ == "__main__":
    main()
