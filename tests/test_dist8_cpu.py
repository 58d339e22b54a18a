"""World size 8 on CPU (gloo, oracle backend): the dealings BASELINE configs[2] / [3] / [4] make on one 8-GPU node.

configs[2] is 8 ranks x batch shard with SyncBN over the global batch; configs[3]'s 225 tiles are 15 groups of 16 dealt to 8 ranks
(2,2,2,2,2,2,2,1); configs[4] is 64 noise realisations over 8 ranks.  Every other multi-process test runs two ranks; the host code
that deals shards, sizes the SyncBN divisor and keys the Philox streams by rank is the same code that runs over RCCL on the GPUs
(engine.trainer.GanEngine / DistSync, api.predict_array, api.predict_ensemble), so these run it at the world size the benchmark uses.
"""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

S, T, B_LOCAL, CIN, NZ, CH, WORLD = 12, 1, 1, 3, 2, 2, 8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(1)
    return dist


def _build(ops):
    from downscaling.engine.networks import DiscriminatorNet, GeneratorNet
    from tests.helpers import randomize
    gen = GeneratorNet(ops, S, CIN, NZ, CH, T, feature_channels=32, seed=5)
    disc = DiscriminatorNet(ops, S, S, CIN, CH, T, feature_channels=8, seed=6)
    return gen, disc, randomize(gen, 21), randomize(disc, 22)


def _data(rank, step):
    g = torch.Generator().manual_seed(300 + 10 * step + rank)
    return (torch.randn(B_LOCAL, T, S, S, CIN, generator=g, dtype=torch.float64),
            torch.randn(B_LOCAL, T, S, S, CH, generator=g, dtype=torch.float64))


def _train_worker(rank, port, outdir):
    dist = _init(rank, port)
    from downscaling.engine.trainer import AdamTF, DistSync, GanEngine, PhiloxSource
    from oracle.torch_backend import TorchOps
    ops = TorchOps()
    gen, disc, _, _ = _build(ops)
    sync = DistSync()
    assert sync.world_size == WORLD and sync.active
    eng = GanEngine(gen, disc, PhiloxSource(ops, seed=99, rank=rank), 0.1, n_critic=2, sync=sync, sync_bn=True)
    g_opt, d_opt = AdamTF(1e-4, 0.5, 0.9, 0.1), AdamTF(4e-4, 0.5, 0.9, 0.1)
    for step in range(2):
        low, high = _data(rank, step)
        logs = eng.train_step(low, high, g_opt, d_opt)
    assert gen.bn1.count == WORLD * B_LOCAL * T * (S // 2) ** 2          # SyncBN divisor: the GLOBAL batch's pixel count
    torch.save({"g": {v.name: v.value.clone() for v in gen.params.vars},
                "d": {v.name: v.value.clone() for v in disc.params.vars}, "seed": eng.noise.seed,
                "logs": {k: float(v) for k, v in logs.items() if v is not None}},
               os.path.join(outdir, f"rank{rank}.pt"))
    dist.destroy_process_group()


class _GlobalDraws:
    def __init__(self, seeds):
        from tests.helpers import Draws
        self.d = [Draws(s, B_LOCAL, T, S, NZ, CH, 0.1) for s in seeds]

    def noise(self):
        return torch.cat([d.noise() for d in self.d], 0)

    def inst(self):
        return torch.cat([d.inst() for d in self.d], 0)

    def eps(self):
        return torch.cat([d.eps() for d in self.d], 0)


@pytest.mark.timeout(900)
def test_eight_rank_step_equals_global_batch_reference():
    """configs[2] at its world size: 8 ranks x 1 sample, SyncBN, two train steps = the single-process restatement on the global
    batch of 8 (weights of both networks, SN / BN state, every logged scalar); the 8 replicas stay bit-identical."""
    from oracle import torch_model as TM
    from oracle.torch_backend import TorchOps
    from tests.helpers import rel_err
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_train_worker, args=(_free_port(), out), nprocs=WORLD, join=True)
        r = [torch.load(os.path.join(out, f"rank{i}.pt")) for i in range(WORLD)]
    assert len({x["seed"] for x in r}) == WORLD                      # rank-keyed noise streams
    for net in ("g", "d"):
        for k in r[0][net]:
            for i in range(1, WORLD):
                assert torch.equal(r[0][net][k], r[i][net][k]), (k, i)
    _, _, gw, dw = _build(TorchOps())
    og, od = TM.AdamTF(1e-4), TM.AdamTF(4e-4)
    draws = _GlobalDraws([x["seed"] for x in r])
    for step in range(2):
        lows, highs = zip(*[_data(rank, step) for rank in range(WORLD)])
        ref_logs = TM.train_step(gw, dw, torch.cat(lows, 0), torch.cat(highs, 0), draws, og, od, n_critic=2)
    for net, w in (("g", gw), ("d", dw)):
        for k in w:
            assert rel_err(r[0][net][k], w[k]) < 1e-7, (net, k)
    for i in range(1, WORLD):
        assert r[i]["logs"] == r[0]["logs"]
    for k in ("g_loss", "g_disc_loss", "d_loss", "d_gradient_pen", "g_gradient_param", "d_gradient_param"):
        assert abs(r[0]["logs"][k] - float(ref_logs[k])) < 1e-7 * max(1.0, abs(float(ref_logs[k]))), k


def _api_small():
    import downscaling.api as api
    from downscaling.engine import runtime
    from oracle.torch_backend import TorchOps
    runtime.set_ops(TorchOps(torch.float64))
    # groups of 16 tiles as api.py:132 (BATCH_SIZE 8 -> group_size 16), small tiles so that the CPU backend finishes
    api.IMG_SIZE, api.SEQUENCE_LENGTH, api.NOISE_CHANNELS, api.BATCH_SIZE = 20, 1, 4, 8
    return api


def _predict_worker(rank, port, outdir):
    dist = _init(rank, port)
    from downscaling.engine.trainer import DistSync
    from downscaling.gan.models import make_generator
    api = _api_small()
    orig = api.make_generator
    api.make_generator = lambda *a, **k: make_generator(*a, feature_channels=32, **k)     # (a narrow generator: CPU time)
    try:
        network = api.get_network(allow_random_init=True, random_seed=11)
    finally:
        api.make_generator = orig
    network.noise_generator.std = 0.0                    # deterministic generator: the blend must not depend on the sharding
    rng = np.random.default_rng(0)
    fields = rng.standard_normal((1, 100, 100, 3)).astype(np.float32)
    fields[..., 2] = fields[..., 2] * 800 + 1500
    plan = api.tile_plan(100, 100, 1, 0.37)
    assert plan["ncols"] == plan["nrows"] == 15          # 225 tiles = 15 groups of 16 (the last one holds a single tile)
    mine = list(range(rank, 15, WORLD))
    assert len(mine) == (2 if rank < 7 else 1)           # (2,2,2,2,2,2,2,1)
    out, cnt = api.predict_array(fields, overlap_factor=0.37, network=network, return_count=True, sync=DistSync())
    np.savez(os.path.join(outdir, f"pred{rank}.npz"), out=out, cnt=cnt)
    if rank == 0:
        o1, c1 = api.predict_array(fields, overlap_factor=0.37, network=network, return_count=True)
        np.savez(os.path.join(outdir, "single.npz"), out=o1, cnt=c1)
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_225_tiles_dealt_to_eight_ranks_equal_single_process():
    """configs[3]'s tile count at configs[2]'s world size: 15 groups of 16 over 8 ranks, per-rank sum / count grids all-reduced
    once — every rank holds the single-process blend."""
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_predict_worker, args=(_free_port(), out), nprocs=WORLD, join=True)
        r = [np.load(os.path.join(out, f"pred{i}.npz")) for i in range(WORLD)]
        single = np.load(os.path.join(out, "single.npz"))
        assert int(single["cnt"].max()) > 1
        for x in r:
            np.testing.assert_array_equal(x["cnt"], single["cnt"])
            np.testing.assert_allclose(x["out"], single["out"], rtol=1e-6, atol=1e-6, equal_nan=True)


def _ensemble_worker(rank, port, outdir):
    dist = _init(rank, port)
    from downscaling.engine.trainer import DistSync
    from downscaling.gan.models import make_generator
    api = _api_small()
    orig = api.make_generator
    api.make_generator = lambda *a, **k: make_generator(*a, feature_channels=32, **k)
    try:
        network = api.get_network(allow_random_init=True, random_seed=11)
    finally:
        api.make_generator = orig
    tiles = torch.randn(8, 1, 20, 20, 3, generator=torch.Generator().manual_seed(4), dtype=torch.float64)
    ens = api.predict_ensemble(tiles, draws=64, network=network, sync=DistSync())
    np.save(os.path.join(outdir, f"ens{rank}.npy"), ens.numpy())
    if rank == 0:
        np.save(os.path.join(outdir, "single.npy"), api.predict_ensemble(tiles, draws=64, network=network).numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_64_member_ensemble_over_eight_ranks_equals_single_process():
    """configs[4] as worded: 64 noise realisations x batch 8 over 8 ranks = 8 members per rank, every member on its own
    (seed, member)-keyed Philox stream: all ranks hold the single-process ensemble member by member, bit for bit."""
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_ensemble_worker, args=(_free_port(), out), nprocs=WORLD, join=True)
        r = [np.load(os.path.join(out, f"ens{i}.npy")) for i in range(WORLD)]
        single = np.load(os.path.join(out, "single.npy"))
    assert single.shape == (64, 8, 1, 20, 20, 2)
    for x in r:
        np.testing.assert_array_equal(x, single)
    assert len({single[m].tobytes() for m in range(64)}) == 64        # 64 distinct realisations
