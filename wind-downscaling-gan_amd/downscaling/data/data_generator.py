"""FlexibleNoiseGenerator with the reference signature
(/root/reference/src/downscaling/data/data_generator.py:319-335), backed by the Philox4x32-10 HIP
kernel (wdg_philox_normal) instead of tf.random.Generator."""
import torch

from downscaling.engine import runtime
from downscaling.engine.trainer import PhiloxSource


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

    def __call__(self, bs=None, channels=None, std=None):
        bs = self.noise_shape[0] if bs is None else int(bs)
        t = self.noise_shape[1]
        x = self.noise_shape[2]
        y = self.noise_shape[3]
        channels = self.noise_shape[4] if channels is None else channels
        std = std or self.std
        ops = runtime.get_ops()
        out = ops.empty(bs, t, x, y, channels)
        self.prng.normal_into(out.view(-1, channels), std)
        return out
