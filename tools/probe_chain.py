import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "wind-downscaling-gan_amd")]
import torch
from downscaling.engine.hipops import ConvGeom, HipOps
ops = HipOps("cuda:0")
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    assert ops.lib.wdg_set_tuning(k.encode(), int(v)) == 0, kv
dev = ops.device
B, T, S, F = 16, 24, 24, 128
g = ConvGeom(3, 3, 1, 1)
w = (torch.randn(3, 3, F, 4 * F, device=dev) * 0.02).contiguous()
pk = ops.pack_weights(w)
gates = torch.randn(T * B, S, S, 4 * F, device=dev) * 0.5
h = torch.zeros(T * B, S, S, F, device=dev)
c = torch.zeros(T * B, S, S, F, device=dev)
small = torch.zeros(1024, device=dev)
def timeit(name, fn, n):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    for _ in range(3): gr.replay()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{name}: {1e3 * ts[4] / n:.2f} us per launch")
def chain_fill():
    for _ in range(T): small.add_(1.0)
def chain_nok():      # epilogue only (t = 0 form: no recurrent convolution), same grid
    for t in range(T):
        sl = slice(t * B, (t + 1) * B)
        ops.convlstm16_step(None, pk, gates[sl], None, c[sl], h[sl], g, F)
def chain_full():
    for t in range(T):
        sl, pv = slice(t * B, (t + 1) * B), slice((t - 1) * B, t * B)
        ops.convlstm16_step(h[pv] if t else None, pk, gates[sl], c[pv] if t else None, c[sl], h[sl], g, F)
def chain_full_1tile():
    for t in range(T):
        sl, pv = slice(t * B, t * B + 1), slice((t - 1) * B, (t - 1) * B + 1)
        ops.convlstm16_step(h[pv] if t else None, pk, gates[sl], c[pv] if t else None, c[sl], h[sl], g, F)
timeit("tiny torch add_ chain", chain_fill, T)
timeit("step, epilogue only (no K loop)", chain_nok, T)
timeit("step, full", chain_full, T)
timeit("step, full, ONE tile (24 workgroups)", chain_full_1tile, T)
