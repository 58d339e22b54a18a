"""`make_generator` / `make_discriminator` with the reference signatures
(/root/reference/src/downscaling/gan/models.py:9-17,76-84) returning Keras-like model objects whose
arithmetic runs on the hand-written HIP kernels (engine.networks on engine.hipops).

Tensors are channels-last (B, T, H, W, C) fp32, as in the reference.  Inputs may be torch tensors
(any device) or numpy arrays; outputs are torch tensors on the GPU (`predict` returns numpy).
"""
import os
from pathlib import Path

import numpy as np
import torch

from downscaling.engine import runtime
from downscaling.engine.networks import DiscriminatorNet, GeneratorNet
from downscaling.engine.common import round4


def _to_dev(x, ops):
    """Dense row-major (B,T,H,W,C) tensor of the backend dtype on the backend device (numpy arrays coming
    out of fancy indexing / np.stack are often not C-contiguous)."""
    if isinstance(x, torch.Tensor):
        return x.to(device=ops.device, dtype=ops.dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x)).to(device=ops.device, dtype=ops.dtype).contiguous()


class _Metrics:
    def __init__(self, metrics=None):
        self.metrics = list(metrics or [])

    def update_state(self, y_true, y_pred, sample_weight=None):
        from downscaling.gan.metrics import pointwise_scope
        with pointwise_scope():       # the sibling metrics share ONE fused pass over (y_true, y_pred); dropped on exit
            for m in self.metrics:
                m.update_state(y_true, y_pred, sample_weight)


class _Model:
    """The slice of keras.Model the reference uses: call / predict / trainable_weights / compile /
    save_weights / load_weights / optimizer / compiled_loss / compiled_metrics / metrics."""

    name = "model"

    def __init__(self, net):
        self.net = net
        self.ops = net.ops
        self.optimizer = None
        self.loss = None
        self.compiled_metrics = _Metrics()
        self._compiled = False

    # -- keras.Model surface -------------------------------------------------------------------------
    def compile(self, optimizer=None, loss=None, metrics=None, **kwargs):
        self.optimizer, self.loss = optimizer, loss
        self.compiled_metrics = _Metrics(metrics)
        self._compiled = True

    def compiled_loss(self, y_true, y_pred, sample_weight=None, regularization_losses=None):
        if self.loss is None:
            raise RuntimeError("compile() with a loss first")
        value = self.loss(y_true, y_pred)
        if sample_weight is not None:
            value = value * torch.as_tensor(sample_weight, dtype=value.dtype, device=value.device).mean()
        for r in regularization_losses or []:
            value = value + r
        return value

    def _assert_compile_was_called(self):
        if not self._compiled:
            raise RuntimeError("You must compile your model before training/testing.")
        return True

    @property
    def metrics(self):
        return self.compiled_metrics.metrics

    @property
    def trainable_weights(self):
        return [v.value for v in self.net.params.trainable]

    @property
    def non_trainable_weights(self):
        return [v.value for v in self.net.params.non_trainable]

    @property
    def weights(self):
        return [v.value for v in self.net.params.vars]

    def count_params(self):
        return sum(v.size for v in self.net.params.vars)

    def get_weights_dict(self):
        return self.net.params.get_weights()

    def set_weights_dict(self, mapping, strict=True):
        return self.net.params.set_weights(mapping, strict=strict)

    def save_weights(self, filepath, overwrite=True, save_format=None, **kwargs):
        """Keras `Model.save_weights(prefix)` of the reference (ganbase.py:132-135) writes the TF tensor-bundle format
        `<prefix>.index` + `<prefix>.data-00000-of-00001`; so does this (keys `<variable>/.ATTRIBUTES/VARIABLE_VALUE`,
        the names of weights-55.ckpt).  `save_format="npz"` (or a path ending in .npz) writes one `<prefix>.npz`."""
        filepath = os.fspath(filepath)
        Path(filepath).parent.mkdir(parents=True, exist_ok=True)
        if save_format == "npz" or filepath.endswith(".npz"):
            base = filepath[:-4] if filepath.endswith(".npz") else filepath
            np.savez(base + ".npz", **{k.replace("/", "|"): v for k, v in self.get_weights_dict().items()})
            return
        if save_format not in (None, "tf"):
            raise ValueError(f"save_format {save_format!r}: only 'tf' (tensor bundle) and 'npz' are available")
        from downscaling.engine.tf_bundle import write_bundle
        write_bundle(filepath, self.get_weights_dict())

    def load_weights(self, filepath, *args, strict=True, **kwargs):
        """Keras `load_weights(prefix)` (ganbase.py:137-140) from `<prefix>.index` + data shards (TF tensor bundle) or
        `<prefix>.npz`.  strict (default): every variable of this model must be in the checkpoint — a foreign or
        partly matching checkpoint raises KeyError instead of leaving randomly initialised weights behind; checkpoint
        keys that name no variable are reported with a warning (the shipped discriminator checkpoint holds
        `layer_with_weights-11..13` of the shortcut variant, for example).  strict=False restores what matches.
        Returns (restored, missing, unused) name lists."""
        import warnings
        filepath = os.fspath(filepath)
        if os.path.exists(filepath + ".npz"):
            with np.load(filepath + ".npz") as z:
                mapping = {k.replace("|", "/"): z[k] for k in z.files}
        elif os.path.exists(filepath + ".index"):
            from downscaling.engine.tf_bundle import read_bundle
            mapping = {k: v for k, v in read_bundle(filepath).items() if not k.startswith(("optimizer", "save_counter"))}
        else:
            raise FileNotFoundError(f"no checkpoint at {filepath}(.npz|.index)")
        restored, missing, unused = self.set_weights_dict(mapping, strict=strict)
        if missing:
            warnings.warn(f"{self.name}.load_weights({filepath!r}): {len(missing)} variable(s) keep their current values: "
                          f"{missing[:6]}{' ...' if len(missing) > 6 else ''}")
        if unused:
            warnings.warn(f"{self.name}.load_weights({filepath!r}): {len(unused)} checkpoint value(s) unused: "
                          f"{unused[:6]}{' ...' if len(unused) > 6 else ''}")
        self.last_load_report = dict(restored=restored, missing=missing, unused=unused)
        return restored, missing, unused


class Generator(_Model):
    name = "generator"
    inference_precision = "fp32"   # "bf16" / "fp16": 16-bit-operand MFMA for the inference forward (BASELINE configs[3] / [4])
    graph_inference = True         # replay the inference forward from a captured HIP graph (GeneratorNet.forward_inference)

    def __call__(self, inputs, training=False, mask=None, precision=None):
        image, noise = inputs
        ops, net = self.ops, self.net
        lazy = getattr(noise, "is_lazy_noise", False)   # data_generator.LazyNoise: drawn straight into the input buffer
        image = _to_dev(image, ops)
        noise = noise if lazy else _to_dev(noise, ops)
        B, T = image.shape[0], image.shape[1]
        assert tuple(image.shape[1:]) == (net.T, net.S, net.S, net.in_channels), image.shape
        assert tuple(noise.shape) == (B, net.T, net.S, net.S, net.noise_channels), noise.shape
        ok = getattr(ops, "input_assemble_ok", None)
        precision = precision or ("fp32" if training else self.inference_precision)
        if lazy and hasattr(noise, "fill_with_image") and image.is_contiguous() and image.dtype == ops.dtype and ok is not None and \
                ok(net.in_channels, net.noise_channels, net.buffers(B)["x0"].shape[-1]):
            rows16 = net.input_rows16(B, precision) if precision in ("bf16", "fp16") and hasattr(net, "input_rows16") else None
            if rows16 is not None:
                # inference precision: the first layer rounds the input to its operand format while staging — assembled in that
                # format it is the same bits and half the bytes, written once
                noise.fill_with_image(rows16, image)
                net.mark_input16(B, precision)
            else:
                noise.fill_with_image(net.input_rows(B), image)  # [image | noise | 0] in one pass, the same Philox stream
        else:
            net.set_image(image)
            if lazy:
                noise.fill(net.noise_view(B))
            else:
                net.set_noise(noise)
        if training or not self.graph_inference:
            out_tm = net.forward(B, bool(training), precision=precision)
        else:
            out_tm = net.forward_inference(B, precision=precision)
        out = torch.empty(B, T, net.S, net.S, net.out_channels, dtype=ops.dtype, device=ops.device)
        net.from_time_major(out_tm, out)
        return out

    call = __call__

    def predict(self, inputs, batch_size=32, precision=None, **kwargs):
        image, noise = inputs
        n = len(image)
        outs = []
        for i in range(0, n, batch_size):
            outs.append(self([image[i:i + batch_size], noise[i:i + batch_size]], training=False,
                             precision=precision).cpu().numpy())
        return np.concatenate(outs, axis=0)


class Discriminator(_Model):
    name = "discriminator"

    def __call__(self, inputs, training=False, mask=None):
        low, high = inputs
        ops, net = self.ops, self.net
        low, high = _to_dev(low, ops), _to_dev(high, ops)
        B, T = low.shape[0], low.shape[1]
        net.set_low(low)
        high_tm = ops.zeros(T * B, net.S, net.S, round4(net.ch))
        net.to_time_major(high, high_tm)
        net.set_high_tm(high_tm, B)
        return net.forward(B, bool(training)).clone().view(B, 1)

    call = __call__


def make_generator(
        image_size: int,
        in_channels: int,
        noise_channels: int,
        out_channels: int,
        n_timesteps: int,
        batch_size: int = None,
        feature_channels=128
):
    return Generator(GeneratorNet(runtime.get_ops(), image_size, in_channels, noise_channels, out_channels,
                                  n_timesteps, feature_channels=feature_channels))


def make_discriminator(
        low_res_size: int,
        high_res_size: int,
        low_res_channels: int,
        high_res_channels: int,
        n_timesteps: int,
        batch_size: int = None,
        feature_channels: int = 16,
        *,
        shortcut_variant: bool = False
):
    """Reference signature (gan/models.py:76-84) plus one keyword-only extension: shortcut_variant=True builds the
    graph the shipped weights-55 discriminator checkpoint was trained with — the split connection of models.py:127-130
    taken after one pass of the `>= 4` loop (the published test `i > 1` can never succeed, so the published code
    never builds it and Keras' lazy restore silently leaves `layer_with_weights-11..13` of the checkpoint unused)."""
    if low_res_size != high_res_size:
        raise NotImplementedError("The discriminator assumes that the low res and high res images have the same size."
                                  "Perhaps you should upsample your low res image first?")
    return Discriminator(DiscriminatorNet(runtime.get_ops(), low_res_size, high_res_size, low_res_channels,
                                          high_res_channels, n_timesteps, feature_channels=feature_channels,
                                          shortcut_variant=shortcut_variant))
