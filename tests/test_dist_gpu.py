"""The N > 1 path on the HIP backend: two processes sharing cuda:0 (gloo rendezvous, gradients / BatchNorm statistics
staged through the host because RCCL refuses two ranks on one device) run GanEngine.train_step on one batch shard
each; replicas must stay bit-identical and equal the fp64 CPU restatement on the concatenated global batch.
(The 8-GPU RCCL run itself is the driver's; this pins the data-parallel logic on the real kernels.)"""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
S, B_LOCAL, CIN, CH, WORLD = 32, 2, 3, 2, 2
# (T, noise channels, generator feature_channels, discriminator feature_channels): a small pair, and the reference's DEFAULT
# widths with a recurrence (models.py:16,83; 20 noise channels: api.py:68) — SyncBN over 128-channel BatchNorms, the split-K
# ConvLSTM gates and the 16-feature fused recurrent step inside a two-rank train step
CASES = {"small": (1, 4, 32, 8), "default_widths": (3, 20, 128, 16)}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(ops, case):
    from downscaling.engine.networks import DiscriminatorNet, GeneratorNet
    from tests.helpers import randomize
    T, NZ, F, FD = CASES[case]
    gen = GeneratorNet(ops, S, CIN, NZ, CH, T, feature_channels=F, seed=5)
    disc = DiscriminatorNet(ops, S, S, CIN, CH, T, feature_channels=FD, seed=6)
    return gen, disc, randomize(gen, 21), randomize(disc, 22)


def _data(rank, step, T):
    g = torch.Generator().manual_seed(100 + 10 * step + rank)
    return (torch.randn(B_LOCAL, T, S, S, CIN, generator=g, dtype=torch.float64),
            torch.randn(B_LOCAL, T, S, S, CH, generator=g, dtype=torch.float64))


def _worker(rank, port, outdir, case):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    for p in (root, root / "wind-downscaling-gan_amd"):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from downscaling.engine.hipops import HipOps
    from downscaling.engine.trainer import AdamTF, DistSync, GanEngine, PhiloxSource
    ops = HipOps("cuda:0")
    gen, disc, _, _ = _build(ops, case)
    eng = GanEngine(gen, disc, PhiloxSource(ops, seed=99, rank=rank), 0.1, n_critic=2, sync=DistSync(), sync_bn=True)
    g_opt, d_opt = AdamTF(1e-4, 0.5, 0.9, 0.1), AdamTF(4e-4, 0.5, 0.9, 0.1)
    for step in range(2):
        low, high = _data(rank, step, CASES[case][0])
        eng.train_step(low.float().to(ops.device), high.float().to(ops.device), g_opt, d_opt)
    torch.cuda.synchronize()
    torch.save({"g": {v.name: v.value.detach().cpu().clone() for v in gen.params.vars},
                "d": {v.name: v.value.detach().cpu().clone() for v in disc.params.vars}, "seed": eng.noise.seed},
               os.path.join(outdir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("case", list(CASES))
def test_two_ranks_on_hip_backend_equal_global_batch_reference(case):
    T, NZ = CASES[case][:2]
    from oracle import torch_model as TM
    from oracle.torch_backend import TorchOps
    from tests.helpers import Draws, rel_err
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_worker, args=(_free_port(), out, case), nprocs=WORLD, join=True)
        r = [torch.load(os.path.join(out, f"rank{i}.pt")) for i in range(WORLD)]
    for net in ("g", "d"):          # replicas bit-identical: deterministic SN + identical all-reduced gradients
        for k in r[0][net]:
            assert torch.equal(r[0][net][k], r[1][net][k]), k

    class GlobalDraws:
        def __init__(self, seeds):
            self.d = [Draws(s, B_LOCAL, T, S, NZ, CH, 0.1) for s in seeds]

        def noise(self):
            return torch.cat([d.noise() for d in self.d], 0)

        def inst(self):
            return torch.cat([d.inst() for d in self.d], 0)

        def eps(self):
            return torch.cat([d.eps() for d in self.d], 0)

    _, _, gw, dw = _build(TorchOps(), case)
    og, od = TM.AdamTF(1e-4), TM.AdamTF(4e-4)
    draws = GlobalDraws([x["seed"] for x in r])
    for step in range(2):
        lows, highs = zip(*[_data(rank, step, T) for rank in range(WORLD)])
        TM.train_step(gw, dw, torch.cat(lows, 0), torch.cat(highs, 0), draws, og, od, n_critic=2)
    for net, w in (("g", gw), ("d", dw)):
        worst = max(rel_err(r[0][net][k], w[k]) for k in w)
        assert worst < 1e-4, (net, worst)
