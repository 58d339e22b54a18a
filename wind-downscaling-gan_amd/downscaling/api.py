"""Inference driver with the reference names and defaults
(/root/reference/src/downscaling/api.py:21-160): constants, get_network, the tiled `predict` (tile plan, per-tile
latitude flip, normalisation, groups of 16, 2-px crop, mean blend) and the `downscale` wrappers around it.

The reference carries its grids in xarray Datasets and reads netCDF / GeoTIFF through xarray + rasterio, none of which
exist in the GPU image.  Here the grids are `downscaling.io.GridDataset` objects (labelled numpy arrays; xarray objects
passed in are converted), the files are read by `downscaling.io` (NetCDF-3 via scipy, .npz, a built-in GeoTIFF reader),
and the array core (`tile_plan`, `predict_array`) runs on the generator's device from upload to blended result."""
import math
import os
from pathlib import Path

import numpy as np

from downscaling.data.data_generator import FlexibleNoiseGenerator, LazyGroupNoise
from downscaling.gan import train, metrics
from downscaling.gan.ganbase import GAN
from downscaling.gan.models import make_generator, make_discriminator
from downscaling.io import GridDataset

WEIGHTS_PATH = (Path(__file__) / '../weights-55.ckpt').resolve()
SEQUENCE_LENGTH = 24
IMG_SIZE = 96
BATCH_SIZE = 8
NOISE_CHANNELS = 20
NOISE_STD = 0.1
NB_INPUTS = 3
NB_OUTPUTS = 2
# api.py:47-48: the high-resolution template has 26 x / 18 x the ERA5 points per latitude / longitude
UPSAMPLING = {'latitude': 26, 'longitude': 18}

__all__ = ['WEIGHTS_PATH', 'SEQUENCE_LENGTH', 'IMG_SIZE', 'BATCH_SIZE', 'NOISE_CHANNELS', 'NOISE_STD', 'NB_INPUTS',
           'NB_OUTPUTS', 'process_topo', 'process_era5', 'build_high_res_template_from_era5', 'get_network',
           'predict', 'downscale', 'tile_plan', 'predict_array', 'predict_ensemble', 'GAN', 'make_generator', 'make_discriminator',
           'FlexibleNoiseGenerator', 'GridDataset']


def _axis_name(names, *prefixes):
    """First name starting with one of the prefixes (the reference finds 'lon_1' / 'lat_1' / 'x' / 'y' this way)."""
    for n in names:
        if str(n).startswith(prefixes):
            return n
    raise KeyError(f"no coordinate starting with {prefixes} among {list(names)}")


def build_high_res_template_from_era5(ds_era5, range_lon=None, range_lat=None):
    """api.py:46-62: the target grid.  Per axis: the requested range (or the whole ERA5 extent) is covered by
    UPSAMPLING[axis] x (number of ERA5 points inside the range) evenly spaced points, end points included; the result
    keeps the dataset's other coordinates (time) and names the new axes 'lon_1' / 'lat_1'."""
    ds = GridDataset.from_xarray(ds_era5)
    axes = {}
    for axis, rng in (('longitude', range_lon), ('latitude', range_lat)):
        coord = ds.coords[axis]
        if rng:
            lo, hi = float(rng[0]), float(rng[1])
            # label slice in the coordinate's own direction: longitudes ascend, ERA5 latitudes descend (api.py:53,57)
            inside = ds.sel_range(axis, lo, hi) if axis == 'longitude' else ds.sel_range(axis, hi, lo)
            count = len(inside.coords[axis])
        else:
            lo, hi, count = float(coord.min()), float(coord.max()), len(coord)
        axes[axis] = np.linspace(lo, hi, UPSAMPLING[axis] * count)
    coords = {k: v for k, v in ds.coords.items() if k not in axes}
    coords.update(lon_1=axes['longitude'], lat_1=axes['latitude'])
    return GridDataset(coords)


def process_era5(ds_era5, high_res_template):
    """api.py:40-43: the 10 m wind components at the ERA5 grid point nearest to every template point."""
    ds, tpl = GridDataset.from_xarray(ds_era5), GridDataset.from_xarray(high_res_template)
    lon, lat = _axis_name(tpl.coords, 'lon'), _axis_name(tpl.coords, 'lat')
    winds = GridDataset(ds.coords, {v: ds.variables[v] for v in ('u10', 'v10')})
    return winds.sel_nearest(rename={'longitude': lon, 'latitude': lat}, longitude=tpl.coords[lon], latitude=tpl.coords[lat])


def process_topo(raster_topo, high_res_template):
    """api.py:31-37: band 0 of the DEM raster, as 'elevation', at the raster pixel nearest to every template point."""
    dem, tpl = GridDataset.from_xarray(raster_topo), GridDataset.from_xarray(high_res_template)
    lon, lat = _axis_name(tpl.coords, 'lon'), _axis_name(tpl.coords, 'lat')
    name = next(iter(dem.variables))
    dims, arr = dem.variables[name]
    if 'band' in dims:
        arr = np.take(arr, 0, axis=dims.index('band'))
        dims = tuple(d for d in dims if d != 'band')
    band0 = GridDataset({k: dem.coords[k] for k in dims}, {'elevation': (dims, arr)})
    return band0.sel_nearest(rename={'x': lon, 'y': lat}, x=tpl.coords[lon], y=tpl.coords[lat])


# (metric constructors, in the order api.py:77-81 lists them)
_GENERATOR_METRICS = ('AngularCosineDistance', 'LogSpectralDistance', 'WeightedRMSEForExtremes', 'WindSpeedWeightedRMSE',
                      'SpatialKS')


def get_network(weights_path=WEIGHTS_PATH, allow_random_init=None, random_seed=None):
    """The shipped network (api.py:65-86): generator and discriminator on 96 x 96 tiles of 24 time steps, 3 inputs
    (u10, v10, elevation), 20 noise channels of std 0.1, compiled with the Adam pair of gan/train.py and the five
    generator metrics, then restored from weights-55.ckpt.
    The checkpoint blobs are absent from the reference tree (.MISSING_LARGE_BLOBS): a missing checkpoint raises, as the
    reference's load_weights would, unless `allow_random_init` (or DOWNSCALING_ALLOW_RANDOM_INIT=1) is set."""
    print('Loading network...')
    if random_seed is None and os.environ.get('DOWNSCALING_RANDOM_SEED'):
        random_seed = int(os.environ['DOWNSCALING_RANDOM_SEED'])      # reproducible CLI runs (the reference's are not)
    tile = dict(n_timesteps=SEQUENCE_LENGTH)
    generator = make_generator(IMG_SIZE, NB_INPUTS, NOISE_CHANNELS, NB_OUTPUTS, **tile)
    discriminator = make_discriminator(IMG_SIZE, IMG_SIZE, NB_INPUTS, NB_OUTPUTS, **tile)
    noise = FlexibleNoiseGenerator((BATCH_SIZE, SEQUENCE_LENGTH, IMG_SIZE, IMG_SIZE, NOISE_CHANNELS), std=NOISE_STD,
                                   random_seed=random_seed)
    gan = GAN(generator, discriminator, noise_generator=noise)
    gan.compile(generator_optimizer=train.generator_optimizer(),
                discriminator_optimizer=train.discriminator_optimizer(),
                generator_metrics=[getattr(metrics, name)() for name in _GENERATOR_METRICS],
                discriminator_loss=train.discriminator_loss,
                metrics=[metrics.discriminator_score_fake(), metrics.discriminator_score_real()])
    if allow_random_init is None:
        allow_random_init = os.environ.get('DOWNSCALING_ALLOW_RANDOM_INIT', '0') == '1'
    try:
        # only the generator is needed for inference; the shipped discriminator checkpoint was trained
        # with the (unreachable) shortcut variant and does not match the published graph
        generator.load_weights(Path(weights_path) / 'generator')
    except FileNotFoundError:
        if not allow_random_init:
            raise
        print(f'WARNING: no checkpoint at {weights_path}; using randomly initialised weights')
    return gan


def _axis_tiles(pixels, overlap_factor):
    """One axis of the tile plan (api.py:101-116): number of IMG_SIZE-wide tiles, their common spacing, the pixels
    that spacing leaves uncovered, and the start offsets.  The count moves from 'just enough to cover the axis'
    (overlap_factor 0) to 'one tile per pixel offset' (overlap_factor 1) with the square of the factor; the leftover
    pixels are absorbed by shifting the first `leftovers` gaps by one pixel each."""
    fewest, most = math.ceil(pixels / IMG_SIZE), pixels - IMG_SIZE
    count = math.floor(fewest + overlap_factor ** 2 * (most - fewest))
    spacing = (pixels - IMG_SIZE) // (count - 1)
    leftovers = pixels - ((count - 1) * spacing + IMG_SIZE)
    if count - leftovers - 1 < 0:
        raise ValueError('negative dimensions are not allowed')       # what np.zeros raises in the reference
    starts = [i * spacing + min(i, leftovers) for i in range(count)]
    return count, spacing, leftovers, starts


def tile_plan(pixels_lat, pixels_lon, time_window, overlap_factor=0.05):
    """The integer tile planner of predict (api.py:98-116).  Only the longitude axis is checked for being wide
    enough: the reference repeats the column test where it means to test the rows (api.py:105), so a too-short
    latitude axis runs into the arithmetic instead — kept, the golden plans in tests/golden/tile_plan.json pin it."""
    if pixels_lon - IMG_SIZE < math.ceil(pixels_lon / IMG_SIZE):
        raise RuntimeError(f'Lon dimension too small: got {pixels_lon} pixels, need at least {IMG_SIZE}')
    assert 0 <= overlap_factor <= 1, 'overlap_factor must be in [0,1] range'
    ncols, xdist, leftovers_x, slices_start_x = _axis_tiles(pixels_lon, overlap_factor)
    nrows, ydist, leftovers_y, slices_start_y = _axis_tiles(pixels_lat, overlap_factor)
    return dict(ntimeseq=time_window // SEQUENCE_LENGTH, ncols=ncols, nrows=nrows, xdist=xdist, ydist=ydist,
                leftovers_x=leftovers_x, leftovers_y=leftovers_y, slices_start_x=slices_start_x,
                slices_start_y=slices_start_y)


def _tile_lat_index(sy):
    """Row indices of one tile: latitude flipped; the sy == 0 tile covers rows 1..96 (api.py:119)."""
    if sy != 0:
        return np.arange(sy + IMG_SIZE - 1, sy - 1, -1)
    return np.arange(IMG_SIZE, 0, -1)


def groups_per_forward(n_groups):
    """How many of a rank's groups of 16 tiles share one forward pass: WDG_PREDICT_GROUPS when set, else the count in 2..5 that
    leaves the fewest padding groups in the last launch (the larger on ties) — 15 groups run as 3 x 5, 8 as 2 x 4, 7 as 2 x 4
    with one padding group.  Measured on the shipped generator (bf16, per group of 16): 2.57 ms alone, 2.37 in pairs, 2.30 in
    fours (profiles/r06x_ab_weight_lds_dma_neutral.txt): the 24 recurrent steps are a dependent chain of launches that 16 tiles
    cannot fill the chip with."""
    env = os.environ.get('WDG_PREDICT_GROUPS')
    if env:
        return max(1, min(int(env), max(1, n_groups)))
    if n_groups <= 5:
        return max(1, n_groups)
    return min(range(2, 6), key=lambda p: ((-n_groups) % p, -p))


def predict_array(fields, overlap_factor=0.05, network=None, return_count=False, sync=None, timings=None, out=None):
    """Array core of predict (api.py:96-151).  fields: (time, lat, lon, 3) float array with channels
    [u10, v10, elevation in metres].  Returns (ntimeseq*24, lat, lon, 2) with NaN where no tile
    contributes (the reference's dataframe simply has no such rows).

    The whole driver runs on the generator's device: the field is uploaded once, tiles are gathered there (latitude
    flip and the sy == 0 off-by-one included), the nanmean / nanstd normalisation over axes (0, 1, 2) is a device
    reduction (accumulated in fp64), every group of 16 tiles goes through the generator without leaving HBM, and the
    2-pixel-cropped tiles are summed / counted into the output grid there; one download at the end.  (The reference
    does this part with numpy / pandas on the host: 3.3 of the 3.5 s of a 1200 x 1200 x 24 h field.)

    `out` (optional): a caller-owned float32 host array / CPU tensor of the result's shape to download into — page-locked
    (`torch.empty(..., pin_memory=True)`) it takes the 276 MB of a 1200 x 1200 x 24 h result at the link's rate instead of the
    ~8 GB/s of a fresh pageable array (34 -> ~8 ms); the same array is returned.

    `sync` (engine.trainer.DistSync, one process per GPU): tiles are independent, so the groups of 16 are dealt
    round-robin to the ranks and the per-rank sum / count grids are all-reduced once at the end — no exchange
    inside the model (SURVEY §8 e).  Every rank returns the full blended field.

    `timings` (dict): filled with the seconds of each phase (upload, tiles + normalisation, generator + blend, mean +
    download), each closed by a device synchronisation — measurement runs only."""
    import time
    import torch

    def lap(name):
        if timings is not None:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            now = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + now - lap.t0
            timings.setdefault('laps', []).append((name, round(now - lap.t0, 5)))
            lap.t0 = now
    lap.t0 = time.perf_counter()
    out_host = out
    network = network or get_network()
    gen = network.generator
    ops = gen.ops
    dev, dt = getattr(ops, "device", "cpu"), ops.dtype
    f = torch.as_tensor(np.asarray(fields, dtype=np.float32)).to(dev)
    f[..., 2] = f[..., 2] / 1e3                                                    # api.py:96
    lap('upload')
    time_window, pixels_lat, pixels_lon = f.shape[:3]
    plan = tile_plan(pixels_lat, pixels_lon, time_window, overlap_factor)
    keys = [(sx, sy, k) for sx in plan['slices_start_x'] for sy in plan['slices_start_y'] for k in range(plan['ntimeseq'])]
    print(f'Applying model to {len(keys)} patches')
    lat_ok = pixels_lat > IMG_SIZE or all(sy != 0 for sy in plan['slices_start_y'])
    if not lat_ok:
        raise RuntimeError('the sy == 0 tile needs lat row 96 (reference slice(IMG_SIZE, 0, -1)): lat dimension too small')
    native_tiles = hasattr(ops, "tiles_gather_normalise") and f.is_cuda and IMG_SIZE * f.shape[3] <= 512
    if native_tiles:
        # csrc/tiling.hip: one gather pass with the NaN-aware sums, one normalisation pass (keys: first column, first = highest
        # row of the flipped tile, sequence index)
        keys4 = torch.tensor([[sx, int(_tile_lat_index(sy)[0]), k, 0] for (sx, sy, k) in keys], dtype=torch.int32, device=dev)
        tensors = ops.tiles_gather_normalise(f.contiguous(), keys4, SEQUENCE_LENGTH, IMG_SIZE).to(dt)
    else:
        rows = {sy: torch.as_tensor(_tile_lat_index(sy).copy(), device=dev) for sy in plan['slices_start_y']}
        tensors = torch.stack([f[k * SEQUENCE_LENGTH:(k + 1) * SEQUENCE_LENGTH].index_select(1, rows[sy])[:, :, sx:sx + IMG_SIZE]
                               for (sx, sy, k) in keys], dim=0)                         # (N, T, H, W, C)
        # nanmean / nanstd over axes (0, 1, 2), keepdims: one statistic per (lon index inside the tile, channel) — api.py:126-129
        valid = ~torch.isnan(tensors)
        n_valid = valid.sum(dim=(0, 1, 2), keepdim=True).double()
        t64 = torch.where(valid, tensors, torch.zeros((), dtype=tensors.dtype, device=dev)).double()
        mean = t64.sum(dim=(0, 1, 2), keepdim=True) / n_valid
        var = (torch.where(valid, t64 - mean, torch.zeros((), dtype=torch.float64, device=dev)) ** 2).sum(dim=(0, 1, 2), keepdim=True) / n_valid
        del t64
        tensors = ((tensors - mean.to(tensors.dtype)) / var.sqrt().to(tensors.dtype)).to(dt)
    nt = plan['ntimeseq'] * SEQUENCE_LENGTH
    acc = torch.zeros(nt, pixels_lat, pixels_lon, NB_OUTPUTS, dtype=torch.float64, device=dev)
    cnt = torch.zeros(nt, pixels_lat, pixels_lon, dtype=torch.int32, device=dev)
    group_size = BATCH_SIZE * 2
    num_groups = math.ceil(tensors.shape[0] / group_size)
    lap('tiles_and_normalisation')
    rank, world = (sync.rank, sync.world_size) if sync is not None else (0, 1)
    # The groups of 16 tiles (api.py:132) are the unit of the reference's NOISE DRAWS, not of the launches here: `per` of this
    # rank's groups share one forward pass — every kernel of the pass carries `per` times the rows for the same weight traffic,
    # and the 24 recurrent steps (a dependent chain of launches that 16 tiles cannot fill the chip with) run once for all of
    # them.  Each group still takes its own draw from the generator's stream, in group order (LazyGroupNoise), and inference
    # treats every tile independently, so the result is the one of group-by-group calls (groups_per_forward: WDG_PREDICT_GROUPS=1).
    mine = list(range(rank, num_groups, world))
    per = groups_per_forward(len(mine))
    with torch.no_grad():
        for c0 in range(0, len(mine), per):
            chunk = mine[c0:c0 + per]
            parts, kparts, n_real = [], [], []
            for t in chunk:
                tensor = tensors[t * group_size:(t + 1) * group_size, ...]
                n_real.append(tensor.shape[0])
                if tensor.shape[0] < group_size and num_groups > 1:
                    # a short last group runs at the resident batch size (zero tiles behind the real ones, their outputs unused):
                    # inference treats every tile independently, and the generator keeps its buffers, plans and graph
                    tensor = torch.cat([tensor, tensor.new_zeros((group_size - tensor.shape[0],) + tuple(tensor.shape[1:]))], dim=0)
                parts.append(tensor)
            gsz = parts[0].shape[0]                                                # 16, or the tile count of a single short group
            if len(parts) < per:
                # a short last launch runs at the resident batch size too: zero groups behind the real ones, no draws for them
                parts += [parts[0].new_zeros(parts[0].shape)] * (per - len(parts))
            tensor = parts[0] if per == 1 else torch.cat(parts, dim=0)
            # fresh noise per group (api.py:136), drawn by the generator model straight into its input buffer
            # (stream order (time, tile, x, y, channel) for the batch size of the group's call, the groups one after the other)
            if per == 1:
                noise = network.noise_generator.lazy(bs=gsz, channels=NOISE_CHANNELS)
            else:
                noise = LazyGroupNoise(network.noise_generator, len(chunk), gsz, network.noise_generator.noise_shape, NOISE_CHANNELS,
                                       network.noise_generator.std, pad_groups=per)
            lap('noise')
            pred_all = gen([tensor, noise])                                        # stays on the device (api.py:137)
            lap('generator')
            for j, t in enumerate(chunk):
                pred = pred_all[j * gsz:(j + 1) * gsz]
                if native_tiles and pred.dtype == torch.float32:
                    ops.tiles_blend(pred.contiguous(), keys4[t * group_size:(t + 1) * group_size].contiguous(), n_real[j], acc, cnt, 2)
                else:
                    for i, (sx, sy, k) in enumerate(keys[t * group_size:(t + 1) * group_size]):
                        r = _tile_lat_index(sy)[2:-2]                              # api.py:148: descending, contiguous
                        ts = slice(k * SEQUENCE_LENGTH, (k + 1) * SEQUENCE_LENGTH)
                        rs, cs = slice(int(r[-1]), int(r[0]) + 1), slice(sx + 2, sx + IMG_SIZE - 2)
                        acc[ts, rs, cs] += pred[i][:, 2:-2, 2:-2].flip(1).double()
                        cnt[ts, rs, cs] += 1
                lap('blend')
                print(f'Predicted {(t + 1) / num_groups:.0%}')
    cnt2d = np.zeros((plan['ntimeseq'], pixels_lat, pixels_lon), dtype=np.int32)   # every rank: the global count (one map per sequence)
    for (sx, sy, k) in keys:
        r = _tile_lat_index(sy)[2:-2]
        cnt2d[k, int(r[-1]):int(r[0]) + 1, sx + 2:sx + IMG_SIZE - 2] += 1
    cnt_host = np.broadcast_to(cnt2d[:, None], (plan['ntimeseq'], SEQUENCE_LENGTH, pixels_lat, pixels_lon)).reshape(nt, pixels_lat, pixels_lon)
    if sync is not None and sync.active:
        sync.all_reduce_sum(acc)
        sync.all_reduce_sum(cnt)
    out = (acc / cnt[..., None].double()).float()                                  # api.py:149-150 (uniform mean); 0/0 -> NaN
    lap('mean')
    if out_host is not None:
        host = out_host if torch.is_tensor(out_host) else torch.from_numpy(out_host)
        if tuple(host.shape) != tuple(out.shape) or host.dtype != torch.float32 or not host.is_contiguous():
            raise ValueError(f"predict_array: `out` must be a contiguous float32 array of shape {tuple(out.shape)}")
        host.copy_(out)
        out = out_host.numpy() if torch.is_tensor(out_host) else out_host
    else:
        out = out.cpu().numpy()
    cnt = cnt_host                                                                 # (the same integer bookkeeping: the count grid is not downloaded)
    lap('download')
    return (out, cnt) if return_count else out


def predict_ensemble(tiles, draws, network=None, sync=None, precision=None, seed=None):
    """Stochastic ensemble inference (BASELINE configs[4]: "64 noise realisations x batch 8"): `draws` realisations of the
    generator on the SAME normalised tiles (N, 24, 96, 96, 3), each with its own noise field — the loop of api.py:132-138
    repeated per realisation.  Returns a (draws, N, 24, 96, 96, 2) tensor on the generator's device.

    Every realisation m draws from its OWN Philox stream, keyed by (seed, m) — not by the process that happens to compute
    it — so member m is the same tensor whatever the number of ranks.  `sync` (engine.trainer.DistSync, one process per
    GPU): realisations are independent, so they are dealt round-robin to the ranks (m % world == rank) with no exchange
    inside the model, and the members are combined by ONE all-reduce of the zero-initialised result at the end (SURVEY §8 e);
    every rank returns the full ensemble.  seed: base seed of the member streams (default: the network's noise
    generator's `random_seed`; an unseeded generator takes a fresh one, agreed between the ranks).
    precision: "fp32" | "bf16" | "fp16" for this call (default: the generator's `inference_precision`)."""
    import torch
    network = network or get_network()
    gen = network.generator
    ops = gen.ops
    dev = getattr(ops, "device", "cpu")
    tiles = torch.as_tensor(tiles).to(device=dev, dtype=ops.dtype)
    if tiles.dim() != 5 or tuple(tiles.shape[1:]) != (SEQUENCE_LENGTH, IMG_SIZE, IMG_SIZE, NB_INPUTS):
        raise ValueError(f'tiles must be (N, {SEQUENCE_LENGTH}, {IMG_SIZE}, {IMG_SIZE}, {NB_INPUTS}), got {tuple(tiles.shape)}')
    draws, n = int(draws), tiles.shape[0]
    if draws < 1:
        raise ValueError('draws must be >= 1')
    rank, world = (sync.rank, sync.world_size) if sync is not None else (0, 1)
    noise_gen = network.noise_generator
    if seed is None:
        seed = noise_gen.random_seed
    if seed is None:
        base = torch.zeros(1, dtype=torch.int64)
        if rank == 0:
            base[0] = int.from_bytes(os.urandom(7), 'little')
        if sync is not None and sync.active:
            sync.all_reduce_sum(base)                              # the other ranks contribute 0: rank 0's draw for everyone
        seed = int(base[0])
    out = torch.zeros(draws, n, SEQUENCE_LENGTH, IMG_SIZE, IMG_SIZE, NB_OUTPUTS, dtype=ops.dtype, device=dev)
    group_size = BATCH_SIZE * 2
    kwargs = {} if precision is None else {'precision': precision}
    # Few tiles, many realisations (configs[4]: 8 tiles x 64): a forward of 8 tiles is a chain of latency-bound launches (24
    # recurrent steps, ~330 kernels), so several MEMBERS share one forward — batch slots [j n, (j + 1) n) hold the tiles with
    # member j's noise (LazyMemberNoise: each member's own stream from offset 0, exactly as alone).  Every forward runs at the
    # same batch size (a short last chunk repeats its last member), so a member's values depend neither on the other members
    # of its launch nor on the number of ranks.
    per = max(1, int(os.environ.get('WDG_ENSEMBLE_TILES', '64')) // n) if n <= group_size else 1
    mine = list(range(rank, draws, world))
    if per > 1 and mine:
        from .data.data_generator import LazyMemberNoise
        stacked = tiles.repeat(per, 1, 1, 1, 1)
        with torch.no_grad():
            for c0 in range(0, len(mine), per):
                chunk = mine[c0:c0 + per]
                members = [FlexibleNoiseGenerator(noise_gen.noise_shape, std=noise_gen.std, random_seed=seed, rank=m)
                           for m in chunk + [chunk[-1]] * (per - len(chunk))]
                y = gen([stacked, LazyMemberNoise(members, n, noise_gen.noise_shape, NOISE_CHANNELS, noise_gen.std)], **kwargs)
                for j, m in enumerate(chunk):
                    out[m] = y[j * n:(j + 1) * n]
        mine = []
    with torch.no_grad():
        for m in mine:
            member = FlexibleNoiseGenerator(noise_gen.noise_shape, std=noise_gen.std, random_seed=seed, rank=m)
            for g0 in range(0, n, group_size):
                group = tiles[g0:g0 + group_size]
                out[m, g0:g0 + group.shape[0]] = gen([group, member.lazy(bs=group.shape[0], channels=NOISE_CHANNELS)], **kwargs)
    if sync is not None and sync.active:
        sync.all_reduce_sum(out)
    return out


def predict(inputs_era5, inputs_topo, high_res_template, overlap_factor=0.05, network=None):
    """api.py:89-152 on GridDatasets: winds (time, lat, lon) + elevation (lat, lon) on the template grid -> downscaled
    u10 / v10 on the pixels at least one tile covers (the reference's dataframe group-by has no rows elsewhere)."""
    era, topo, tpl = (GridDataset.from_xarray(d) for d in (inputs_era5, inputs_topo, high_res_template))
    lat = _axis_name(tpl.dims, 'lat', 'y')
    lon = _axis_name(tpl.dims, 'lon', 'x')
    nt = len(era.coords['time'])
    elevation = np.broadcast_to(topo.transposed('elevation', lat, lon)[None], (nt, len(tpl.coords[lat]), len(tpl.coords[lon])))
    fields = np.stack([era.transposed('u10', 'time', lat, lon), era.transposed('v10', 'time', lat, lon), elevation], axis=-1)
    out, cnt = predict_array(fields, overlap_factor=overlap_factor, network=network, return_count=True)
    keep_lat, keep_lon = cnt[0].any(axis=1), cnt[0].any(axis=0)
    out = out[:, keep_lat][:, :, keep_lon]
    coords = {'time': era.coords['time'][:out.shape[0]], lat: tpl.coords[lat][keep_lat], lon: tpl.coords[lon][keep_lon]}
    return GridDataset(coords, {v: (('time', lat, lon), out[..., i]) for i, v in enumerate(('u10', 'v10'))})


def downscale(era5, raster_topo, range_lon=None, range_lat=None, overlap_factor=0.05, network=None):
    """api.py:155-160: template from the ERA5 grid, nearest-neighbour regridding of winds and DEM onto it, tiled
    generator inference.  `era5` needs `longitude` (ascending), `latitude` (descending, as ERA5), `time`, `u10`, `v10`;
    `raster_topo` is a (band, y, x) raster as `downscaling.io.open_raster` / `xr.open_rasterio` return it."""
    template = build_high_res_template_from_era5(era5, range_lon=range_lon, range_lat=range_lat)
    return predict(process_era5(era5, template), process_topo(raster_topo, template), template,
                   overlap_factor=overlap_factor, network=network)
