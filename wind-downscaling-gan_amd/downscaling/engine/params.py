"""Flat parameter storage.

All trainable variables of a network live in ONE flat fp32 buffer (and their gradients / Adam slots
in parallel flat buffers): the optimizer is a single fused kernel launch and the data-parallel
gradient exchange is a single RCCL all-reduce over `grads`.  Variable order mirrors Keras'
`model.trainable_weights` order of the reference graphs (gan/models.py:9-142), and every variable
keeps its TF name/shape (`layer_with_weights-N/...` keys decoded from weights-55.ckpt).
"""
import math

import numpy as np
import torch


class Var:
    """One TF variable.  `tf_shape` is its checkpoint shape; `shape` is the stored shape, which differs only for a variable
    with a `gap = (axis, pos, width)`: `width` extra entries at index `pos` along `axis` that are zero and stay zero (their
    gradient is a product with an all-zero alignment channel, and Adam maps a zero gradient on a zero slot to a zero
    update).  Used for the kernel of a layer that reads a channel concatenation whose second segment starts at a padded,
    16-byte aligned channel (GeneratorNet: feature_channels % 16 == 8)."""
    __slots__ = ("name", "shape", "tf_shape", "gap", "trainable", "init", "offset", "size", "tf_size", "value", "grad",
                 "value_pad", "grad_pad", "fresh", "lazy_ok")

    def __init__(self, name, shape, trainable, init, gap=None):
        self.name, self.tf_shape, self.trainable, self.init = name, tuple(shape), trainable, init
        self.gap = gap
        st = list(shape)
        if gap is not None:
            st[gap[0]] += gap[2]
        self.shape = tuple(st)
        self.tf_size = int(np.prod(shape))
        self.size = int(np.prod(self.shape))
        self.offset = -1
        self.value = None
        self.grad = None
        self.value_pad = self.grad_pad = None    # flat views up to the 4-element alignment slot (zeros beyond `size`)
        self.fresh = False                       # ParamStore.zero_grad(lazy=True): the gradient slot is UNDEFINED until first written
        self.lazy_ok = False                     # ... only for variables whose gradient writer honours `fresh` (layers.Conv kernels)


class ParamStore:
    def __init__(self, ops):
        self.ops = ops
        self.vars = []
        self.version = 0  # bumped whenever values change outside SN (optimizer step, load, set)
        self.flat = self.grads = self.state = None

    def touch(self):
        """Values (flat / state) were written in place by something other than set_weights / the optimizer: everything derived
        from them and cached on `version` (packed kernel layouts, 16-bit copies, inference BatchNorm affines, captured inference
        graphs) is stale."""
        self.version += 1

    def copy_from(self, other):
        """This store's variables <- another store of the same graph (the trainer's twin discriminator)."""
        self.flat.copy_(other.flat)
        self.state.copy_(other.state)
        self.touch()

    def add(self, name, shape, init, trainable=True, gap=None):
        v = Var(name, shape, trainable, init, gap)
        self.vars.append(v)
        return v

    @staticmethod
    def expand(v, a):
        """TF-shaped numpy array -> stored shape (zeros in the gap)."""
        a = np.asarray(a)
        if v.gap is None:
            return a
        axis, pos, width = v.gap
        z = np.zeros(a.shape[:axis] + (width,) + a.shape[axis + 1:], dtype=a.dtype)
        return np.concatenate([np.take(a, range(pos), axis=axis), z, np.take(a, range(pos, a.shape[axis]), axis=axis)], axis=axis)

    @staticmethod
    def squeeze(v, t):
        """stored-shape tensor (value or gradient) -> TF shape."""
        if v.gap is None:
            return t
        axis, pos, width = v.gap
        return torch.cat([t.narrow(axis, 0, pos), t.narrow(axis, pos + width, t.shape[axis] - pos - width)], dim=axis)

    def finalize(self, rng):
        """Allocate the flat buffers and initialise every variable (Keras default initialisers)."""
        train = [v for v in self.vars if v.trainable]
        other = [v for v in self.vars if not v.trainable]
        # 4-element alignment keeps every weight view 16-byte aligned for the b128 loads
        off = 0
        for v in train:
            v.offset = off
            off += (v.size + 3) // 4 * 4
        self.n_train = off
        host = np.zeros(off, dtype=np.float64)
        for v in train:
            host[v.offset:v.offset + v.size] = self.expand(v, v.init(v.tf_shape, rng)).reshape(-1)
        self.flat = self.ops.from_host(host)
        self.grads = self.ops.zeros(off)
        off2 = 0
        for v in other:
            v.offset = off2
            off2 += (v.size + 3) // 4 * 4
        host2 = np.zeros(max(off2, 4), dtype=np.float64)
        for v in other:
            host2[v.offset:v.offset + v.size] = self.expand(v, v.init(v.tf_shape, rng)).reshape(-1)
        self.state = self.ops.from_host(host2)
        for v in train:
            v.value = self.flat[v.offset:v.offset + v.size].view(v.shape)
            v.grad = self.grads[v.offset:v.offset + v.size].view(v.shape)
            v.value_pad = self.flat[v.offset:v.offset + (v.size + 3) // 4 * 4]
            v.grad_pad = self.grads[v.offset:v.offset + (v.size + 3) // 4 * 4]
        for v in other:
            v.value = self.state[v.offset:v.offset + v.size].view(v.shape)
            v.value_pad = self.state[v.offset:v.offset + (v.size + 3) // 4 * 4]
        self.trainable = train
        self.non_trainable = other
        # segment table for the g/d_gradient_param metric (mean over variables of mean(g^2))
        pairs = []
        for v in train:
            pairs += [v.offset, v.offset + v.size]
        self.seg_pairs = torch.tensor(pairs, dtype=torch.int64).to(self.flat.device)
        self.seg_out = self.ops.empty(len(train))
        # mean(g^2) is over the TF shape: a gapped variable's stored mean is rescaled (1.0 everywhere else)
        self.seg_scale = self.ops.from_host(np.array([v.size / v.tf_size for v in train], dtype=np.float64)) if any(
            v.gap is not None for v in train) else None

    LAZY_MIN = 1 << 18      # variables of >= 1 MiB take part in lazy zeroing

    def zero_grad(self, lazy=False):
        """grads <- 0.  lazy=True (the trainer's critic / generator updates, whose next backward pass writes every kernel
        gradient): the big convolution kernels — 94 % of the discriminator's 34 MB — are not filled but marked `fresh`; the first
        weight-gradient launch of such a variable then STORES instead of accumulating (layers.Conv.backward_weights), and
        `settle()` zero-fills whatever is still fresh before anything reads the buffer as a whole."""
        if not lazy:
            self.grads.zero_()
            for v in self.trainable:
                v.fresh = False
            return
        if getattr(self, "_lazy_ranges", None) is None:
            big = [v for v in self.trainable if v.lazy_ok and v.size >= self.LAZY_MIN]
            ranges, pos = [], 0
            for v in big:
                if v.offset > pos:
                    ranges.append((pos, v.offset))
                pos = v.offset + (v.size + 3) // 4 * 4
            if pos < self.n_train:
                ranges.append((pos, self.n_train))
            self._lazy_big, self._lazy_ranges = big, ranges
        if hasattr(self.ops, "zero_ranges") and self._lazy_ranges:
            self.ops.zero_ranges(self.grads, self._lazy_ranges)          # (one launch instead of one fill per range)
        else:
            for a, b in self._lazy_ranges:
                self.grads[a:b].zero_()
        for v in self._lazy_big:
            v.fresh = True

    def settle(self):
        """Zero-fills every gradient slot still marked fresh (a pass that did not form that weight gradient): call before the
        flat gradient buffer is read as a whole (optimizer step, all-reduce, gradient-norm metric, sums of two networks)."""
        for v in getattr(self, "_lazy_big", None) or ():
            if v.fresh:
                v.grad.zero_()
                v.fresh = False

    def set_grads_sum(self, a, b):
        """grads <- a.grads + b.grads (both settled): the weight gradients of two passes that ran on copies of this network."""
        torch.add(a.grads, b.grads, out=self.grads)
        for v in self.trainable:
            v.fresh = False

    def num_trainable(self):
        return sum(v.tf_size for v in self.trainable)

    def by_name(self, name):
        for v in self.vars:
            if v.name == name:
                return v
        raise KeyError(name)

    # ---- host <-> device ------------------------------------------------------------------------
    def get_weights(self):
        """{name: numpy array} of every variable (trainable and not), TF shapes."""
        return {v.name: self.squeeze(v, v.value).detach().double().cpu().numpy().astype(np.float32) for v in self.vars}

    def set_weights(self, mapping, strict=True):
        """Copies `mapping[name]` into every variable it names.  Returns (restored, missing, unused): variable names
        that were set, variables of this network absent from the mapping, and mapping keys that name no variable.
        strict=True raises KeyError when anything is missing (BEFORE touching any value); shape clashes always raise."""
        names = {v.name for v in self.vars}
        missing = [v.name for v in self.vars if v.name not in mapping]
        unused = [k for k in mapping if k not in names]
        if strict and missing:
            raise KeyError(f"{len(missing)} variable(s) not in the checkpoint / mapping: {missing[:6]}"
                           + (" ..." if len(missing) > 6 else ""))
        for v in self.vars:
            if v.name in mapping and tuple(np.asarray(mapping[v.name]).shape) != v.tf_shape:
                raise ValueError(f"{v.name}: shape {np.asarray(mapping[v.name]).shape} != {v.tf_shape}")
        restored = []
        for v in self.vars:
            if v.name in mapping:
                v.value.copy_(self.ops.from_host(self.expand(v, np.asarray(mapping[v.name]))).view(v.shape))
                restored.append(v.name)
        self.version += 1
        return restored, missing, unused


# ---- Keras default initialisers -------------------------------------------------------------------
def glorot_uniform(fan_in, fan_out):
    def init(shape, rng):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, size=shape)
    return init


def conv_glorot(shape, rng):
    kh, kw, cin, cout = shape
    lim = math.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
    return rng.uniform(-lim, lim, size=shape)


def orthogonal(shape, rng):
    rows = int(np.prod(shape[:-1]))
    cols = shape[-1]
    a = rng.standard_normal((max(rows, cols), min(rows, cols)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    if rows < cols:
        q = q.T
    return q.reshape(shape)


def zeros_init(shape, rng):
    return np.zeros(shape)


def ones_init(shape, rng):
    return np.ones(shape)


def lstm_bias(F):
    def init(shape, rng):  # Keras unit_forget_bias: gate order i,f,c,o
        b = np.zeros(shape)
        b[F:2 * F] = 1.0
        return b
    return init


def sn_u_init(shape, rng):  # tf.initializers.TruncatedNormal(stddev=0.02)
    x = rng.standard_normal(shape) * 0.02
    bad = np.abs(x) > 0.04
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum())) * 0.02
        bad = np.abs(x) > 0.04
    return x
