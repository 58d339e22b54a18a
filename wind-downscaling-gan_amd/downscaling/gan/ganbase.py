"""`GAN` with the reference constructor / compile / train_step / test_step / call / save / load
surface (/root/reference/src/downscaling/gan/ganbase.py:8-140).  The step itself is
engine.trainer.GanEngine on the HIP kernels."""
import os
from pathlib import Path

import numpy as np
import torch

from downscaling.engine.trainer import DistSync, GanEngine
from downscaling.gan import train as _train
from downscaling.gan.models import _Metrics, _to_dev


def unpack_x_y_sample_weight(data):
    if not isinstance(data, (tuple, list)):
        return data, None, None
    if len(data) == 1:
        return data[0], None, None
    if len(data) == 2:
        return data[0], data[1], None
    return data[0], data[1], data[2]


class GAN:
    def __init__(self, generator, discriminator, noise_generator, n_critic=3, reconstruction_loss=None,
                 distributed=None, sync_bn=True, *args, **kwargs):
        self.generator = generator
        self.discriminator = discriminator
        self.noise_generator = noise_generator
        self.reconstruction_loss = reconstruction_loss
        self._n_critic = n_critic
        self.compiled_metrics = _Metrics()
        self._compiled = False
        sync = None
        if distributed is None:
            import torch.distributed as dist
            distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if distributed:
            sync = DistSync()
            # every rank must draw its OWN generator noise / gradient-penalty eps / instance noise: key the Philox stream
            # of a seeded generator with the rank (an unseeded one takes a fresh os.urandom seed per process)
            if hasattr(noise_generator, "set_rank"):
                noise_generator.set_rank(sync.rank)
        self.engine = GanEngine(generator.net, discriminator.net, noise_generator.prng, noise_generator.std,
                                n_critic=n_critic, sync=sync, sync_bn=sync_bn)

    def _assert_compile_was_called(self):
        return self.generator._assert_compile_was_called() and self.discriminator._assert_compile_was_called()

    @property
    def metrics(self):
        return self.compiled_metrics.metrics

    def compile(self,
                generator_optimizer,
                discriminator_optimizer,
                generator_loss=None,
                generator_metrics=None,
                discriminator_loss=None,
                metrics=None,
                **kwargs):
        self.compiled_metrics = _Metrics(metrics)
        self._compiled = True
        # any callable loss(real_output, fake_output) -> scalar (ganbase.py:44-45 goes through compiled_loss); the reference's
        # own Wasserstein form takes the step's built-in path, anything else the coupled path (GanEngine._critic_coupled)
        builtin = (None, _train.discriminator_loss, _train.discriminator_adversarial_loss)
        self._d_loss_fn = None if discriminator_loss in builtin else discriminator_loss
        self.generator.compile(generator_optimizer, generator_loss, metrics=generator_metrics)
        self.discriminator.compile(discriminator_optimizer, discriminator_loss or _train.discriminator_loss)

    def train_step(self, data):
        low_res, high_res, sample_weight = unpack_x_y_sample_weight(data)
        self._assert_compile_was_called()
        ops = self.generator.ops
        low_res, high_res = _to_dev(low_res, ops), _to_dev(high_res, ops)
        res = self.engine.train_step(low_res, high_res, self.generator.optimizer, self.discriminator.optimizer,
                                     sample_weight=sample_weight, reconstruction_loss=self.reconstruction_loss,
                                     d_loss_fn=getattr(self, "_d_loss_fn", None))
        return_metrics = {k: res[k] for k in ('g_loss', 'g_disc_loss', 'g_reco_loss', 'd_loss', 'd_gradient_pen',
                                              'g_gradient_param', 'd_gradient_param')}
        if self.generator.metrics or self.metrics:
            B, T = low_res.shape[0], low_res.shape[1]
            gen = self.generator.net
            fake = torch.empty(B, T, gen.S, gen.S, gen.out_channels, dtype=ops.dtype, device=ops.device)
            gen.from_time_major(self.engine.last_fake_tm, fake)
            self.generator.compiled_metrics.update_state(high_res, fake, sample_weight)
            self.compiled_metrics.update_state(res['_d_real'].reshape(1), res['_d_fake'].reshape(1), None)
        for metric in self.metrics:
            return_metrics[metric.name] = metric.result()
        for metric in self.generator.metrics:
            return_metrics[f'g_{metric.name}'] = metric.result()
        return return_metrics

    def test_step(self, data):
        x, y, sample_weight = unpack_x_y_sample_weight(data)
        ops = self.generator.ops
        res = self.engine.test_step(_to_dev(x, ops), _to_dev(y, ops), d_loss_fn=getattr(self, "_d_loss_fn", None))
        return_metrics = {'loss': res['loss']}
        for metric in self.metrics:
            return_metrics[metric.name] = metric.result()
        return return_metrics

    def call(self, inputs, training=None, mask=None):
        low_res, high_res, sample_weight = unpack_x_y_sample_weight(inputs)
        batch_size = low_res.shape[0]
        noise = self.noise_generator(batch_size)
        return self.generator([low_res, noise], training=bool(training))

    __call__ = call

    def fit(self, data, epochs=1, steps_per_epoch=None, verbose=1):
        """Minimal stand-in for keras.Model.fit: `data` is an iterable of (low_res, high_res[, sw])."""
        history = []
        for epoch in range(epochs):
            for i, batch in enumerate(data):
                if steps_per_epoch is not None and i >= steps_per_epoch:
                    break
                logs = self.train_step(batch)
                history.append({k: (float(v) if v is not None else None) for k, v in logs.items()})
                if verbose:
                    print(f"epoch {epoch} step {i}: " + " ".join(f"{k}={v:.4g}" for k, v in history[-1].items() if v is not None))
        return history

    def save_weights(self, filepath, *args, **kwargs):
        self.generator.save_weights(os.path.join(filepath, 'generator'), *args, **kwargs)
        self.discriminator.save_weights(os.path.join(filepath, 'discriminator'), *args, **kwargs)

    def load_weights(self,
                     filepath,
                     *args, **kwargs):
        self.generator.load_weights(Path(filepath) / f'generator', *args, **kwargs)
        self.discriminator.load_weights(Path(filepath) / f'discriminator', *args, **kwargs)
