"""`AutoEncoder` of the reference (/root/reference/src/downscaling/autoencoder/autoencoder.py:11-51) as far as the GAN
path uses it: its ENCODER is the `feature_extractor` of `reconstruction_loss` (gan/train.py:19-26, fed from
autoencoder/features_encoding.py:10-19).  The encoder runs on the HIP kernels (engine.networks.EncoderNet) and is
differentiable w.r.t. its input through torch autograd, which is what GAN.train_step needs (ganbase.py:57-61).  The
decoder and the autoencoder's own training are outside the GAN hot path and not built.
"""
import torch

from downscaling.engine import runtime
from downscaling.engine.networks import EncoderNet
from downscaling.gan.models import _Model, _to_dev


class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model):
        ctx.model = model
        ctx.token = model._bump()
        return model.net.forward(x)

    @staticmethod
    def backward(ctx, grad):
        if ctx.token != ctx.model._token:
            raise RuntimeError("encoder: backward after a later forward (its activations were overwritten); "
                               "evaluate the branch that needs a gradient last")
        return ctx.model.net.backward_input(grad.contiguous()), None


class Encoder(_Model):
    name = "encoder"

    def __init__(self, net):
        super().__init__(net)
        self._token = 0

    def _bump(self):
        self._token += 1
        return self._token

    def __call__(self, inputs, training=False, mask=None):
        """inputs (B, T, S, S, 2) -> (B, T, latent).  Differentiable w.r.t. `inputs` (weights frozen: inference mode)."""
        if training:
            raise NotImplementedError("the encoder is built as a frozen feature extractor (inference mode)")
        x = _to_dev(inputs, self.ops)
        if isinstance(inputs, torch.Tensor) and inputs.requires_grad:
            return _EncodeFn.apply(x, self)
        self._bump()
        return self.net.forward(x)

    def predict(self, inputs, **kwargs):
        return self(inputs).detach().cpu().numpy()


class AutoEncoder:
    """AutoEncoder(img_size, time_steps, latent_dimension, batch_size): `.encoder` as in the reference; the decoder
    (autoencoder.py:38-51) is not part of the GAN path."""

    def __init__(self, img_size, time_steps, latent_dimension, batch_size=None):
        self.latent_dimension = latent_dimension
        self.img_size = img_size
        self.time_steps = time_steps
        self.batch_size = batch_size
        self.encoder = self.make_encoder()

    def make_encoder(self):
        return Encoder(EncoderNet(runtime.get_ops(), self.img_size, self.time_steps, self.latent_dimension))

    def make_decoder(self):
        raise NotImplementedError("the decoder is outside the GAN hot path (SURVEY 8 f4 covers the encoder)")

    @property
    def decoder(self):
        return self.make_decoder()

    def load_weights(self, filepath, *args, **kwargs):
        """Restores the encoder's variables from a checkpoint of the whole autoencoder (keys prefixed `encoder/`) or
        of the encoder alone."""
        import os
        filepath = os.fspath(filepath)
        if os.path.exists(filepath + ".index"):
            from downscaling.engine.tf_bundle import read_bundle
            w = read_bundle(filepath)
            pref = "encoder/"
            sub = {k[len(pref):]: v for k, v in w.items() if k.startswith(pref)}
            self.encoder.set_weights_dict(sub or w, strict=False)
            return
        self.encoder.load_weights(filepath)
