import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch
from downscaling.engine.hipops import ConvGeom, HipOps
ops = HipOps("cuda:0")
B=32
x = torch.randn(B,128,128,160, device=ops.device)
w = (torch.randn(5,5,16,160, device=ops.device)*0.05).contiguous()
bias = torch.randn(16, device=ops.device)
pk = ops.pack_weights(w)
g = ConvGeom(5,5,1,2)
ys={}
for mode in (False, True, False, True):
    ops.upconv_colfwd = mode
    y = torch.empty(B,256,256,16, device=ops.device)
    ts=[]
    for r in range(8):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); ops.upconv_fwd(x, pk, bias, y, g, act=True); e1.record(); torch.cuda.synchronize()
        if r>1: ts.append(e0.elapsed_time(e1))
    ys[mode]=y
    print("column" if mode else "composite", round(sorted(ts)[len(ts)//2],3), "ms")
print("maxdiff", float((ys[True]-ys[False]).abs().max()), float(ys[False].abs().max()))
z = ops._scratch("upc_col", B,128,128,400)
ts=[]
for r in range(8):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); ops.lib.wdg_upconv_gather(z.data_ptr(), bias.data_ptr(), y.data_ptr(), 16, 256*256*16, B, 128, 128, 16, 1, 0.2, ops.stream); e1.record(); torch.cuda.synchronize()
    if r>1: ts.append(e0.elapsed_time(e1))
print("gather alone", round(sorted(ts)[len(ts)//2],3), "ms")
