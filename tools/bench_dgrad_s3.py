"""Times the discriminator's 7x7 stride-3 32 -> 64 data gradient + LayerNorm backward (wdg_conv_dgrad_lnbwd) on both routes:
the patch kernel of dgrad_patch_s3.hip and the implicit-GEMM epilogue.   python tools/bench_dgrad_s3.py [n_img] [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wind-downscaling-gan_amd"))
import torch
from downscaling.engine.hipops import HipOps, ConvGeom

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ops = HipOps()
dev = ops.device
H = W = 256
g = ConvGeom(7, 7, 3, 1)
Ho = (H + 2 - 7) // 3 + 1
torch.manual_seed(0)
dy = torch.randn(n, Ho, Ho, 64, device=dev)
w = torch.randn(7, 7, 32, 64, device=dev) / 56.0
pk = ops.pack_weights(w)
y = torch.randn(n, H, W, 16, device=dev)
mr = torch.stack([y.mean(-1).reshape(-1), 1.0 / torch.sqrt(y.var(-1, unbiased=False).reshape(-1) + 1e-3)], 1).contiguous()
gamma = torch.rand(16, device=dev) + 0.5
dg, db, dbias = (torch.zeros(16, device=dev) for _ in range(3))
ws = ops.lnbwd_scratch(16)
dx = ops.zeros(n, H, W, 32)
flops = 2.0 * 49 * 32 * 64 * n * Ho * Ho
warm = False
routes = [int(v) for v in os.environ.get("WDG_S3_ROUTES", "1,0").split(",")]      # 1 + 2 * DBG: skeletons of a -DWDG_S3_SKELETONS build
for route in routes:
    assert ops.lib.wdg_set_tuning(b"dgrad_s3", route) == 0
    for par in ((True, False) if route < 2 else ((True,) if ((route >> 1) & 127) in (16, 48) else (False,))):
        args = (dg, db, dbias, ws) if par else (None, None, None, None)
        for _ in range(3 if warm else 60):          # (the first case also warms the clocks up: cold, a 470 us launch measures 530)
            ops.conv_dgrad_lnbwd(dy, pk, dx, g, y, mr, gamma, 16, 16, 0.2, *args)
        warm = True
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        for _ in range(reps):
            ops.conv_dgrad_lnbwd(dy, pk, dx, g, y, mr, gamma, 16, 16, 0.2, *args)
        e1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        if route and hasattr(ops.lib, "wdg_s3_prof") and os.environ.get("WDG_S3_PROF"):
            import ctypes as C
            ops.lib.wdg_s3_prof.restype = C.c_int
            ops.lib.wdg_s3_prof.argtypes = [C.c_void_p, C.c_int]
            assert ops.lib.wdg_s3_prof(None, 1) == 0
            ops.conv_dgrad_lnbwd(dy, pk, dx, g, y, mr, gamma, 16, 16, 0.2, *args)
            torch.cuda.synchronize()
            out = (C.c_ulonglong * 8)()
            assert ops.lib.wdg_s3_prof(out, 0) == 0
            names = ["prologue+patch+barrier", "first weights wait", "DMA issue", "first tap of class", "other taps", "epilogue", "drain"]
            tot = sum(out[k] for k in range(7))
            print("   phase clocks per wave (s_memtime ticks), %d waves:" % out[7], "  ".join("%s %.1f (%.1f%%)" % (names[k], out[k] / max(1, out[7]), 100.0 * out[k] / max(1, tot)) for k in range(7)))
        print(f"route={('patch_s3' if route == 1 else 'patch_s3 skeleton %d' % (route >> 1)) if route else 'igemm'} param_grads={par}: {us:8.1f} us  {flops / us / 1e6:6.1f} TFLOP/s  frac of 157.3 = {flops / us / 1e6 / 157.3:.3f}")
