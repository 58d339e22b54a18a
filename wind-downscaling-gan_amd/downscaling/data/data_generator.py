"""FlexibleNoiseGenerator with the reference signature
(/root/reference/src/downscaling/data/data_generator.py:319-335), backed by the Philox4x32-10 HIP
kernel (wdg_philox_normal) instead of tf.random.Generator."""
import datetime as _dt
from pathlib import Path as _Path

import numpy as np
import torch

from downscaling.engine import runtime
from downscaling.engine.trainer import PhiloxSource


class LazyNoise(object):
    """A pending FlexibleNoiseGenerator draw (see FlexibleNoiseGenerator.lazy): `shape` is the (batch, time, x, y, channels)
    shape the tensor would have; `fill(view2d)` writes the draw into a [time * batch * x * y, channels] view."""

    is_lazy_noise = True

    def __init__(self, generator, shape, std):
        self.generator, self.shape, self.std = generator, tuple(shape), std

    def fill(self, view2d):
        assert view2d.shape[0] * view2d.shape[1] == int(np.prod(self.shape))
        self.generator.prng.normal_into(view2d, self.std)

    def fill_with_image(self, rows, image):
        """The same draw, written together with the image into whole rows [time * batch * x * y, ld] of the generator's input
        buffer ([image | noise | zero alignment channels]; one kernel instead of a channel copy and a noise pass)."""
        B, T, X, Y, C = self.shape
        assert rows.shape[0] == T * B * X * Y and tuple(image.shape[:4]) == (B, T, X, Y)
        prng = self.generator.prng
        prng.assemble_at(image, rows, B, X * Y, C, self.std, prng.reserve(rows.shape[0] * C))


class LazyMemberNoise(object):
    """Pending draws of SEVERAL FlexibleNoiseGenerators for one forward pass of batch len(generators) * tiles: batch slots
    [j * tiles, (j + 1) * tiles) take generator j's stream exactly as a forward of those `tiles` alone would — element order
    (time, tile, x, y, channel) from Philox offset 0 — so a member's noise does not depend on which other members share its
    launch (api.predict_ensemble batches ensemble members this way)."""

    is_lazy_noise = True

    def __init__(self, generators, tiles, noise_shape, channels, std):
        self.generators, self.tiles, self.std = list(generators), int(tiles), std
        self.shape = (len(self.generators) * self.tiles, noise_shape[1], noise_shape[2], noise_shape[3], channels)

    def fill(self, view2d):
        B, T, X, Y, C = self.shape
        assert view2d.shape[0] == T * B * X * Y and view2d.shape[1] == C
        rows = self.tiles * X * Y                       # one timestep of one member: a contiguous row block of the time-major view
        assert (rows * C) % 4 == 0                      # whole Philox blocks per timestep
        for j, g in enumerate(self.generators):
            for t in range(T):
                r0 = (t * B + j * self.tiles) * X * Y
                g.prng.normal_at(view2d[r0:r0 + rows], self.std, t * (rows * C // 4))

    def fill_with_image(self, rows_all, image):
        """As fill, with the image written in the same pass (see LazyNoise.fill_with_image)."""
        B, T, X, Y, C = self.shape
        assert rows_all.shape[0] == T * B * X * Y and tuple(image.shape[:4]) == (B, T, X, Y)
        rows = self.tiles * X * Y
        assert (rows * C) % 4 == 0
        for j, g in enumerate(self.generators):
            img_j = image[j * self.tiles:(j + 1) * self.tiles]
            for t in range(T):
                r0 = (t * B + j * self.tiles) * X * Y
                g.prng.assemble_at(img_j[:, t:t + 1], rows_all[r0:r0 + rows], self.tiles, X * Y, C, self.std, t * (rows * C // 4))


class LazyGroupNoise(object):
    """Pending draws of ONE FlexibleNoiseGenerator for `groups` consecutive calls of `tiles` tiles each, run as one forward pass of
    batch groups * tiles: batch slots [g * tiles, (g + 1) * tiles) take exactly the values the g-th of `groups` successive
    `lazy(bs=tiles)` draws would have had — element order (time, tile, x, y, channel) inside a group, the groups one after the
    other in the stream — so api.predict's result does not depend on how many of its groups of 16 (api.py:132) share a launch."""

    is_lazy_noise = True

    def __init__(self, generator, groups, tiles, noise_shape, channels, std, pad_groups=None):
        """pad_groups >= groups: the forward pass runs pad_groups * tiles batch slots; the slots behind the real groups get no
        draw (their outputs are unused) and the generator's stream does not advance for them."""
        self.generator, self.groups, self.tiles, self.std = generator, int(groups), int(tiles), std
        self.slots = max(self.groups, int(pad_groups or 0))
        self.shape = (self.slots * self.tiles, noise_shape[1], noise_shape[2], noise_shape[3], channels)

    def _walk(self, fn):
        B, T, X, Y, C = self.shape
        rows = self.tiles * X * Y                       # one timestep of one group: a contiguous row block of the time-major view
        assert (rows * C) % 4 == 0                      # whole Philox blocks per timestep
        prng = self.generator.prng
        for g in range(self.groups):
            base = prng.reserve(T * rows * C)           # the group's place in the stream (= one lazy(bs=tiles) draw)
            for t in range(T):
                fn(g, t, (t * B + g * self.tiles) * X * Y, rows, base + t * (rows * C // 4))

    def fill(self, view2d):
        B, T, X, Y, C = self.shape
        assert view2d.shape[0] == T * B * X * Y and view2d.shape[1] == C
        self._walk(lambda g, t, r0, rows, off: self.generator.prng.normal_at(view2d[r0:r0 + rows], self.std, off))

    def fill_with_image(self, rows_all, image):
        """As fill, with the image written in the same pass (see LazyNoise.fill_with_image)."""
        B, T, X, Y, C = self.shape
        assert rows_all.shape[0] == T * B * X * Y and tuple(image.shape[:4]) == (B, T, X, Y)
        n, prng = self.tiles, self.generator.prng
        assert (n * X * Y * C) % 4 == 0
        for g in range(self.groups):
            prng.assemble_slots_at(image[g * n:(g + 1) * n], rows_all, n, X * Y, C, self.std, prng.reserve(T * n * X * Y * C), B, g * n)


class FlexibleNoiseGenerator(object):
    def __init__(self, noise_shape, std=1, random_seed=None, rank=0):
        self.noise_shape = noise_shape
        self.random_seed = random_seed
        self.rank = rank
        self._prng = None
        self.std = std

    @property
    def prng(self):
        if self._prng is None:
            self._prng = PhiloxSource(runtime.get_ops(), self.random_seed, self.rank)
        return self._prng

    def set_rank(self, rank):
        """Data-parallel rank of this process (GAN passes its DistSync rank): decorrelates the ranks' streams of one
        `random_seed`.  Re-keys an already created, still unused source; refuses to re-key one that has been drawn from."""
        rank = int(rank)
        if rank == self.rank:
            return
        if self._prng is not None and self._prng.offset != 0:
            raise RuntimeError("FlexibleNoiseGenerator.set_rank after noise has been drawn: the stream cannot be re-keyed")
        self.rank = rank
        if self._prng is not None and self.random_seed is not None:
            self._prng = PhiloxSource(self._prng.ops, self.random_seed, rank)

    def lazy(self, bs=None, channels=None, std=None):
        """The same draw as __call__, deferred: the generator model that receives the returned LazyNoise writes the Philox
        stream straight into its (time-major) input buffer instead of copying a (batch, time, x, y, channels) tensor there.
        Element order of the stream: (time, batch, x, y, channel) — as in training (engine.trainer) — so the values of a
        given tile depend on the batch size of the call.  Extension of this build; the reference passes tensors."""
        bs = self.noise_shape[0] if bs is None else int(bs)
        channels = self.noise_shape[4] if channels is None else channels
        return LazyNoise(self, (bs, self.noise_shape[1], self.noise_shape[2], self.noise_shape[3], channels),
                         std or self.std)       # same rule as __call__ (reference line 334: a falsy std selects self.std)

    def __call__(self, bs=None, channels=None, std=None):
        bs = self.noise_shape[0] if bs is None else int(bs)
        t = self.noise_shape[1]
        x = self.noise_shape[2]
        y = self.noise_shape[3]
        channels = self.noise_shape[4] if channels is None else channels
        std = std or self.std        # as the reference (data_generator.py:334): std=0 / None both mean "the generator's std"
        ops = runtime.get_ops()
        out = ops.empty(bs, t, x, y, channels)
        self.prng.normal_into(out.view(-1, channels), std)
        return out


# ----------------------------------------------------------------------------------------------------
# Host-side data feeding of the reference (data/data_generator.py:21-293, 296-316, 338-417), restated on
# plain numpy containers: xarray / netCDF4 / s3cmd are not part of this image, so a "day" is a mapping
# {variable: array[time, x, y]} served by a provider instead of an xarray Dataset opened from a file.
# Sampling semantics follow the reference exactly (what is cropped, in which order the random draws are
# made, what the decoder sees, the flip / rot90 augmentation), so a user of the reference finds the same
# classes with the same constructor arguments.
# ----------------------------------------------------------------------------------------------------
class Provider(object):
    """data_generator.py:21-33: `available_dates` + `provide(date)` (a context manager there; here it returns the
    day's {variable: array[time, x, y]} mapping directly)."""

    @property
    def available_dates(self):
        raise NotImplementedError

    def provide(self, date):
        raise NotImplementedError


class ArrayProvider(Provider):
    """In-memory provider: {date: {variable: array[time, x, y]}}."""

    def __init__(self, days):
        self.days = dict(days)

    @property
    def available_dates(self):
        return list(self.days)

    def provide(self, date):
        return self.days[date]


class LocalFileProvider(Provider):
    """data_generator.py:36-60 analogue: one file per day under `root`, named by `pattern`
    (default 'x_{date:%Y%m%d}.npz'; the reference's are netCDF 'x_{date:%Y%m%d}.nc').  `.npz` files hold one
    array[time, x, y] per variable."""

    def __init__(self, root, pattern="x_{date:%Y%m%d}.npz"):
        self.root, self.pattern = _Path(root), pattern

    @property
    def available_dates(self):
        out = []
        for f in sorted(self.root.iterdir()):
            d = self._parse(f.name)
            if d is not None:
                out.append(d)
        return out

    def _parse(self, name):
        # invert `pattern` for the one supported field, {date:%Y%m%d}
        head, _, tail = self.pattern.partition("{date:%Y%m%d}")
        if not (name.startswith(head) and name.endswith(tail)) or len(name) != len(head) + 8 + len(tail):
            return None
        try:
            return _dt.datetime.strptime(name[len(head):len(head) + 8], "%Y%m%d")
        except ValueError:
            return None

    def provide(self, date):
        with np.load(self.root / self.pattern.format(date=date)) as z:
            return {k: z[k] for k in z.files}


class NaiveDecoder(object):
    """data_generator.py:338-360.  Statistics over axes (0, 1, 2) with keepdims: for the (time, x, y, channel)
    patches of the batch generator that is one mean / std per channel."""

    def __init__(self, normalize=True):
        self.normalize_input = normalize

    def __call__(self, img):
        return self.normalize(img) if self.normalize_input else img

    def normalize(self, img):
        ax = (0, 1, 2)
        return (img - np.nanmean(img, axis=ax, keepdims=True)) / np.nanstd(img, axis=ax, keepdims=True)

    def normalize_positive(self, img):
        ax = (0, 1, 2)
        lo, hi = np.nanmin(img, axis=ax, keepdims=True), np.nanmax(img, axis=ax, keepdims=True)
        return (img - lo) / (hi - lo)

    def denormalize(self, img):
        return img * np.nanstd(img) + np.nanmean(img)

    def denormalize_positive(self, img):
        return np.nanmin(img) + img * (np.nanmax(img) - np.nanmin(img))


class _RangeDecoder(object):
    """Shared body of WindSpeedDecoder / WindComponentDecoder (data_generator.py:363-417): zeros are missing
    values (-> NaN), values below the range become `below_val`, values above are clipped."""

    def __init__(self, value_range, below_val, normalize):
        self.value_range, self.below_val, self.normalize_output = value_range, below_val, normalize

    def __call__(self, img):
        img = np.asarray(img)
        dec = np.full(img.shape, np.nan, dtype=np.float32)
        valid = img != 0
        dec[valid] = img[valid]
        with np.errstate(invalid="ignore"):
            dec[dec < self.value_range[0]] = self.below_val
        np.clip(dec, None, self.value_range[1], out=dec)
        return self.normalize(dec) if self.normalize_output else dec


class WindSpeedDecoder(_RangeDecoder):
    def __init__(self, value_range=(np.log10(0.1), np.log10(100)), below_val=np.nan, normalize=False):
        super().__init__(value_range, below_val, normalize)

    def normalize(self, img):
        return (img - self.below_val) / (self.value_range[1] - self.below_val)

    def denormalize(self, img, set_nan=True):
        img = img * (self.value_range[1] - self.below_val) + self.below_val
        with np.errstate(invalid="ignore"):
            img[img < self.value_range[0]] = self.below_val
        if set_nan:
            img[img == self.below_val] = np.nan
        return img


class WindComponentDecoder(_RangeDecoder):
    def __init__(self, value_range=(-10, 10), below_val=np.nan, normalize=True):
        super().__init__(value_range, below_val, normalize)

    def normalize(self, img):
        return (img - np.mean(img)) / np.std(img)

    def denormalize(self, img, set_nan=True):
        img = img * np.std(img) + np.mean(img)
        with np.errstate(invalid="ignore"):
            img[img < self.value_range[0]] = self.below_val
        if set_nan:
            img[img == self.below_val] = np.nan
        return img


class _BatchGenerator(object):
    """data_generator.py:145-293.  One batch = `batch_size` random (time, x, y) crops of ONE day; crops use the
    global numpy RNG (as the reference does, `np.random.randint`), the augmentation uses `self.prng`
    (`RandomState`, reseedable through `reset`)."""

    def __init__(self, input_provider, decoder, output_provider=None, start_date=None, end_date=None,
                 sequence_length=6, patch_length_pixel=30, batch_size=16, transform=True,
                 input_variables=('u10', 'v10', 'blh', 'fsr', 'sp', 'z', 'vo', 'd', 'tpi_500', 'ridge_index_norm'),
                 output_variables=('U_10M', 'V_10M')):
        self.insert_random_img_transforms = transform
        self.batch_size, self.decoder = batch_size, decoder
        self.sequence_length, self.patch_length_pixel = sequence_length, patch_length_pixel
        self.input_variables, self.output_variables = list(input_variables), list(output_variables)
        self.input_provider, self.output_provider = input_provider, output_provider
        dates = set(input_provider.available_dates)
        if output_provider is not None:
            dates &= set(output_provider.available_dates)
        if start_date is not None:
            dates = {d for d in dates if _as_datetime(d) >= _as_datetime(start_date)}
        if end_date is not None:
            dates = {d for d in dates if _as_datetime(d) <= _as_datetime(end_date)}
        self.dates = sorted(dates)
        self.reset()

    def reset(self, random_seed=None):
        self.prng = np.random.RandomState(seed=random_seed)
        self.current_date_index = -1

    def next_date(self):
        self.current_date_index = (self.current_date_index + 1) % len(self.dates)
        return self.dates[self.current_date_index]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        pass

    def __iter__(self):
        return self

    def __len__(self):
        return len(self.dates)

    def __getitem__(self, item):
        return self.generate(self.dates[item])

    def __next__(self):
        return self.generate(self.next_date())

    def __call__(self):
        return next(self)

    def get_random_square_sequences_per_day(self, X, Y=None):
        """X, Y: {variable: array[time, x, y]}.  Draw order x, y, time; `elevation` is served in km."""
        any_var = next(iter(X.values()))
        nt, nx, ny = any_var.shape
        rx = np.random.randint(0, nx + 1 - self.patch_length_pixel)
        ry = np.random.randint(0, ny + 1 - self.patch_length_pixel)
        rt = np.random.randint(0, nt + 1 - self.sequence_length)

        def crop(day, variables):
            planes = []
            for v in variables:
                a = np.asarray(day[v])[rt:rt + self.sequence_length, rx:rx + self.patch_length_pixel,
                                        ry:ry + self.patch_length_pixel]
                planes.append(a / 1e3 if v == "elevation" else a)
            return np.stack(planes, axis=-1)

        if Y is not None:
            return crop(X, self.input_variables), crop(Y, self.output_variables)
        return crop(X, self.input_variables)

    def transform_sequence(self, X, Y=None):
        """Random mirror along each spatial axis, then 0-3 quarter turns; X and Y get the same transform."""
        flip_x = bool(self.prng.randint(2))
        if flip_x:
            X = np.flip(X, axis=1)
            Y = None if Y is None else np.flip(Y, axis=1)
        flip_y = bool(self.prng.randint(2))
        if flip_y:
            X = np.flip(X, axis=2)
            Y = None if Y is None else np.flip(Y, axis=2)
        turns = self.prng.randint(4)
        if turns > 0:
            X = np.rot90(X, k=turns, axes=(1, 2))
            Y = None if Y is None else np.rot90(Y, k=turns, axes=(1, 2))
        return X if Y is None else (X, Y)

    def generate(self, date):
        day_in = self.input_provider.provide(date)
        day_out = self.output_provider.provide(date) if self.output_provider is not None else None
        xs, ys = [], []
        for _ in range(self.batch_size):
            if day_out is not None:
                X, Y = self.get_random_square_sequences_per_day(day_in, day_out)
            else:
                X, Y = self.get_random_square_sequences_per_day(day_in), None
            X = self.decoder(X)                         # the decoder sees the input patch only
            if self.insert_random_img_transforms:
                X, Y = (self.transform_sequence(X, Y) if Y is not None else (self.transform_sequence(X), None))
            xs.append(X)
            ys.append(Y)
        if day_out is not None:
            return np.stack(xs, axis=0), np.stack(ys, axis=0)
        return np.stack(xs, axis=0)


def _as_datetime(d):
    if isinstance(d, _dt.datetime):
        return d
    if isinstance(d, _dt.date):
        return _dt.datetime(d.year, d.month, d.day)
    s = str(d)
    for fmt in ("%Y%m%d", "%Y-%m-%d", "%Y-%m-%d %H:%M:%S"):
        try:
            return _dt.datetime.strptime(s, fmt)
        except ValueError:
            pass
    raise ValueError(f"cannot parse date {d!r}")


class BatchGenerator(object):
    """data_generator.py:96-142: the Keras `Sequence` face of `_BatchGenerator` (one item per day; `len` = the
    number of calendar days spanned).  Worker processes (`num_workers > 1`, Keras OrderedEnqueuer in the
    reference) are not provided: the per-day crops are a few MB of numpy slicing."""

    def __init__(self, input_provider, decoder, output_provider=None, start_date=None, end_date=None,
                 sequence_length=6, patch_length_pixel=30, batch_size=16, transform=True,
                 input_variables=('u10', 'v10', 'blh', 'fsr', 'sp', 'z', 'vo', 'd', 'tpi_500', 'ridge_index_norm'),
                 output_variables=('U_10M', 'V_10M'), num_workers=1):
        self.num_workers = num_workers
        self._bg = _BatchGenerator(input_provider, decoder, output_provider, start_date, end_date, sequence_length,
                                   patch_length_pixel, batch_size, transform, input_variables, output_variables)

    def __len__(self):
        stamps = [_as_datetime(d) for d in self._bg.dates]
        return (max(stamps) - min(stamps)).days + 1

    def __getitem__(self, item):
        return self._bg.generate(self._bg.dates[item])

    def __enter__(self):
        return self._bg

    def __exit__(self, *exc):
        pass


class NoiseGenerator(object):
    """data_generator.py:296-316: four structured noise channels (time-, lon-, lat- and lon/lat-varying).
    The reference builds each channel as reshape(repeat(draw, n), (bs, t, x, y)) with `tf.repeat` on the flattened
    draw, i.e. every drawn value fills n CONSECUTIVE elements of the (bs, t, x, y) row-major layout — which is
    constant along the intended axes only for the time-varying channel; the other three are scrambled.  That
    layout is reproduced as written (the draws come from the Philox kernel, not from tf.random)."""

    def __init__(self, noise_shape, std=1., random_seed=None, rank=0):
        self.noise_shape, self.std = noise_shape, std
        self.random_seed, self.rank = random_seed, rank
        self._prng = None

    @property
    def prng(self):
        if self._prng is None:
            self._prng = PhiloxSource(runtime.get_ops(), self.random_seed, self.rank)
        return self._prng

    @staticmethod
    def layout(draw, n, shape):
        """reshape(repeat(draw.flatten(), n), shape) for torch tensors or numpy arrays."""
        if isinstance(draw, np.ndarray):
            return np.repeat(draw.reshape(-1), n).reshape(shape)
        return torch.repeat_interleave(draw.reshape(-1), n).reshape(shape)

    def __call__(self, bs=None):
        bs = self.noise_shape[0] if bs is None else int(bs)
        t, x, y = self.noise_shape[1], self.noise_shape[2], self.noise_shape[3]
        ops = runtime.get_ops()

        def draw(*shape):
            out = ops.empty(int(np.prod(shape)), 1)
            self.prng.normal_into(out, self.std)
            return out.view(*shape)
        shape = (bs, t, x, y)
        chans = [self.layout(draw(bs, t), x * y, shape), self.layout(draw(bs, x), t * y, shape),
                 self.layout(draw(bs, y), t * x, shape), self.layout(draw(bs, x, y), t, shape)]
        return torch.stack(chans, dim=-1)
