"""Shared helpers of the parity tests (test infrastructure)."""
import numpy as np
import torch


def randomize(net, seed, scale=1.0):
    """Replace the Keras-default initial values by generic random ones (so that gamma/beta/bias/moving
    statistics are not the trivial 1/0) and return them as {TF key: float64 torch tensor}."""
    rng = np.random.default_rng(seed)
    vals = {}
    for v in net.params.vars:
        n = v.name
        if n.endswith("gamma"):
            a = rng.uniform(0.5, 1.5, v.tf_shape)
        elif n.endswith("moving_variance"):
            a = rng.uniform(0.5, 2.0, v.tf_shape)
        elif n.endswith("beta") or n.endswith("bias") or n.endswith("moving_mean"):
            a = rng.normal(0, 0.3, v.tf_shape)
        elif n.endswith("sn_u"):
            a = rng.normal(0, 0.02, v.tf_shape)
        else:
            fan_in = int(np.prod(v.tf_shape[:-1]))
            a = rng.normal(0, scale / np.sqrt(fan_in), v.tf_shape)
        vals[n] = a
    net.params.set_weights(vals)
    return {k: torch.tensor(a, dtype=torch.float64) for k, a in vals.items()}


def weights64(net):
    return {v.name: net.params.squeeze(v, v.value).detach().double().cpu().clone() for v in net.params.vars}   # TF shapes


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def grads64(net):
    return {v.name: net.params.squeeze(v, v.grad).detach().double().cpu().clone() for v in net.params.trainable}


class Draws:
    """Replays the Philox stream exactly as engine.trainer.PhiloxSource consumes it and hands the
    oracle train step its random tensors in API layout ([B,T,S,S,C]); the engine writes noise into
    time-major buffers, so element (t,b,h,w,c) of the stream maps to [b,t,h,w,c]."""

    def __init__(self, seed, B, T, S, nz, ch, std):
        from oracle.torch_backend import philox_normal_np, philox_uniform_np
        self._n, self._u = philox_normal_np, philox_uniform_np
        self.seed, self.offset = seed, 0
        self.B, self.T, self.S, self.nz, self.ch, self.std = B, T, S, nz, ch, std

    def _normal(self, C):
        n = self.T * self.B * self.S * self.S * C
        z = self._n(n, self.seed, self.offset) * self.std
        self.offset += (n + 3) // 4
        return torch.tensor(z, dtype=torch.float64).reshape(self.T, self.B, self.S, self.S, C).transpose(0, 1)

    def noise(self):
        return self._normal(self.nz)

    def inst(self):
        return self._normal(self.ch)

    def eps(self):
        u = self._u(self.B, self.seed, self.offset)
        self.offset += (self.B + 3) // 4
        return torch.tensor(u, dtype=torch.float64)
