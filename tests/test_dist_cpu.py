"""The N > 1 path on CPU: two processes (gloo, world_size 2), one batch shard each, running the same
GanEngine host code that runs over RCCL on the GPUs (flat gradient all-reduce + SyncBN statistics
all-reduce), must reproduce the single-process reference step on the concatenated global batch."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

S, T, B_LOCAL, CIN, NZ, CH, WORLD = 12, 2, 1, 3, 2, 2, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(ops):
    from downscaling.engine.networks import DiscriminatorNet, GeneratorNet
    from tests.helpers import randomize
    gen = GeneratorNet(ops, S, CIN, NZ, CH, T, feature_channels=32, seed=5)
    disc = DiscriminatorNet(ops, S, S, CIN, CH, T, feature_channels=8, seed=6)
    return gen, disc, randomize(gen, 21), randomize(disc, 22)


def _data(rank, step):
    g = torch.Generator().manual_seed(100 + 10 * step + rank)
    return (torch.randn(B_LOCAL, T, S, S, CIN, generator=g, dtype=torch.float64),
            torch.randn(B_LOCAL, T, S, S, CH, generator=g, dtype=torch.float64))


def _worker(rank, port, outdir, sync_bn):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(2)
    from downscaling.engine.trainer import AdamTF, DistSync, GanEngine, PhiloxSource
    from oracle.torch_backend import TorchOps
    ops = TorchOps()
    gen, disc, _, _ = _build(ops)
    eng = GanEngine(gen, disc, PhiloxSource(ops, seed=99, rank=rank), 0.1, n_critic=2, sync=DistSync(), sync_bn=sync_bn)
    g_opt, d_opt = AdamTF(1e-4, 0.5, 0.9, 0.1), AdamTF(4e-4, 0.5, 0.9, 0.1)
    for step in range(2):
        low, high = _data(rank, step)
        logs = eng.train_step(low, high, g_opt, d_opt)
    torch.save({"g": {v.name: v.value.clone() for v in gen.params.vars},
                "d": {v.name: v.value.clone() for v in disc.params.vars}, "seed": eng.noise.seed,
                "logs": {k: float(v) for k, v in logs.items() if v is not None}},
               os.path.join(outdir, f"rank{rank}.pt"))
    dist.destroy_process_group()


class _GlobalDraws:
    """Concatenates the per-rank Philox streams along the batch axis, as the global batch sees them."""

    def __init__(self, seeds):
        from tests.helpers import Draws
        self.d = [Draws(s, B_LOCAL, T, S, NZ, CH, 0.1) for s in seeds]

    def noise(self):
        return torch.cat([d.noise() for d in self.d], 0)

    def inst(self):
        return torch.cat([d.inst() for d in self.d], 0)

    def eps(self):
        return torch.cat([d.eps() for d in self.d], 0)


@pytest.mark.timeout(600)
def test_two_rank_step_equals_global_batch_reference():
    from oracle import torch_model as TM
    from oracle.torch_backend import TorchOps
    from tests.helpers import rel_err
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_worker, args=(_free_port(), out, True), nprocs=WORLD, join=True)
        r = [torch.load(os.path.join(out, f"rank{i}.pt")) for i in range(WORLD)]
    # replicas stay bit-identical (deterministic SN + identical all-reduced gradients)
    for net in ("g", "d"):
        for k in r[0][net]:
            assert torch.equal(r[0][net][k], r[1][net][k]), k
    # ... and equal the single-process reference on the global batch
    _, _, gw, dw = _build(TorchOps())
    og, od = TM.AdamTF(1e-4), TM.AdamTF(4e-4)
    draws = _GlobalDraws([x["seed"] for x in r])
    for step in range(2):
        lows, highs = zip(*[_data(rank, step) for rank in range(WORLD)])
        ref_logs = TM.train_step(gw, dw, torch.cat(lows, 0), torch.cat(highs, 0), draws, og, od, n_critic=2)
    for net, w in (("g", gw), ("d", dw)):
        for k in w:
            assert rel_err(r[0][net][k], w[k]) < 1e-7, (net, k)
    # the logged scalars are those of the GLOBAL batch on every rank (one small all-reduce; ganbase.py:75-81)
    assert r[0]["logs"] == r[1]["logs"]
    for k in ("g_loss", "g_disc_loss", "d_loss", "d_gradient_pen", "g_gradient_param", "d_gradient_param"):
        assert abs(r[0]["logs"][k] - float(ref_logs[k])) < 1e-7 * max(1.0, abs(float(ref_logs[k]))), k


def _gan_rank_worker(rank, port, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(2)
    from downscaling.data.data_generator import FlexibleNoiseGenerator
    from downscaling.engine import runtime
    from downscaling.gan.ganbase import GAN
    from downscaling.gan.models import make_discriminator, make_generator
    from oracle.torch_backend import TorchOps
    runtime.set_ops(TorchOps(torch.float64))
    g, d = make_generator(S, CIN, NZ, CH, T, feature_channels=32), make_discriminator(S, S, CIN, CH, T, feature_channels=8)
    # the caller forgot rank= (get_network does): GAN must key the seeded stream with its DistSync rank
    gan = GAN(g, d, FlexibleNoiseGenerator((B_LOCAL, T, S, S, NZ), std=0.1, random_seed=7))
    assert gan.noise_generator.rank == rank
    noise = gan.noise_generator(bs=2)
    torch.save({"noise": noise.clone(), "seed": gan.noise_generator.prng.seed}, os.path.join(outdir, f"noise{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_gan_decorrelates_seeded_noise_across_ranks():
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_gan_rank_worker, args=(_free_port(), out), nprocs=WORLD, join=True)
        r = [torch.load(os.path.join(out, f"noise{i}.pt")) for i in range(WORLD)]
    assert r[0]["seed"] != r[1]["seed"]
    a, b = r[0]["noise"].flatten(), r[1]["noise"].flatten()
    assert not torch.equal(a, b)
    assert abs(float(torch.corrcoef(torch.stack([a, b]))[0, 1])) < 0.2


def _predict_worker(rank, port, outdir):
    import numpy as np
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(2)
    from downscaling.engine import runtime
    from downscaling.engine.trainer import DistSync
    from oracle.torch_backend import TorchOps
    import downscaling.api as api
    runtime.set_ops(TorchOps(torch.float64))
    api.IMG_SIZE, api.SEQUENCE_LENGTH, api.NOISE_CHANNELS, api.BATCH_SIZE = 20, 2, 5, 1
    network = api.get_network(allow_random_init=True, random_seed=11)
    network.noise_generator.std = 0.0                    # deterministic generator: the blend must not depend on the sharding
    rng = np.random.default_rng(0)
    fields = rng.standard_normal((4, 40, 50, 3)).astype(np.float32)
    fields[..., 2] = fields[..., 2] * 800 + 1500
    out, cnt = api.predict_array(fields, overlap_factor=0.3, network=network, return_count=True, sync=DistSync())
    np.savez(os.path.join(outdir, f"pred{rank}.npz"), out=out, cnt=cnt)
    if rank == 0:
        network2 = api.get_network(allow_random_init=True, random_seed=11)
        network2.noise_generator.std = 0.0
        o1, c1 = api.predict_array(fields, overlap_factor=0.3, network=network2, return_count=True)
        np.savez(os.path.join(outdir, "single.npz"), out=o1, cnt=c1)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_tile_sharded_inference_equals_single_process():
    """configs[3]/[4] multi-GPU path: groups of tiles dealt round-robin to the ranks, sum / count grids all-reduced."""
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_predict_worker, args=(_free_port(), out), nprocs=WORLD, join=True)
        r = [np.load(os.path.join(out, f"pred{i}.npz")) for i in range(WORLD)]
        single = np.load(os.path.join(out, "single.npz"))
        for x in r:
            np.testing.assert_array_equal(x["cnt"], single["cnt"])
            np.testing.assert_allclose(x["out"], single["out"], rtol=1e-6, atol=1e-6, equal_nan=True)


def _ensemble_worker(rank, port, outdir):
    import numpy as np
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.set_num_threads(2)
    from downscaling.engine import runtime
    from downscaling.engine.trainer import DistSync
    from oracle.torch_backend import TorchOps
    import downscaling.api as api
    runtime.set_ops(TorchOps(torch.float64))
    api.IMG_SIZE, api.SEQUENCE_LENGTH, api.NOISE_CHANNELS, api.BATCH_SIZE = 20, 2, 5, 1    # groups of 2 tiles
    network = api.get_network(allow_random_init=True, random_seed=11)
    tiles = torch.randn(3, 2, 20, 20, 3, generator=torch.Generator().manual_seed(4), dtype=torch.float64)
    ens = api.predict_ensemble(tiles, draws=5, network=network, sync=DistSync())
    np.save(os.path.join(outdir, f"ens{rank}.npy"), ens.numpy())
    if rank == 0:
        network2 = api.get_network(allow_random_init=True, random_seed=11)
        np.save(os.path.join(outdir, "single.npy"), api.predict_ensemble(tiles, draws=5, network=network2).numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ensemble_sharded_over_ranks_equals_single_process_member_by_member():
    """configs[4] multi-GPU path (api.predict_ensemble): noise realisations dealt round-robin to the ranks, every member on its
    own (seed, member)-keyed Philox stream, one all-reduce of the result — every rank holds the single-process ensemble,
    member by member, and the members differ from each other (5 draws over 2 ranks: an uneven deal, 3 tiles in groups of 2)."""
    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_ensemble_worker, args=(_free_port(), out), nprocs=WORLD, join=True)
        r = [np.load(os.path.join(out, f"ens{i}.npy")) for i in range(WORLD)]
        single = np.load(os.path.join(out, "single.npy"))
    assert single.shape == (5, 3, 2, 20, 20, 2)
    for x in r:
        np.testing.assert_array_equal(x, single)          # same arithmetic on the same streams: bit-identical
    for m in range(1, 5):
        assert np.abs(single[m] - single[0]).max() > 1e-6
