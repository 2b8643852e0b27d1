"""Keras-shaped layer descriptions and ``Sequential`` container.

Stands in for ``tensorflow.keras.layers`` / ``tf.keras.Sequential`` as the reference demos use them
(demo_celeba.py:51-124, demo_mnist.py:48-86): same class names, constructor arguments, ``add``,
``output_shape`` / ``input_shape``, ``count_params``, ``summary``, ``trainable_variables``,
``get_weights`` / ``set_weights``, call with ``training=``.  The layers hold no arithmetic: a
``Sequential`` is compiled into fused stages and executed by ``engine.Net`` on the HIP kernels.

Parameters of one top-level model live in ONE flat float32 HBM buffer (``ParamStore.theta``) so the
optimiser is a single fused Adam launch and data-parallel training is a single all-reduce; each
Keras variable is a view into it, stored exactly in TF's layout (Dense [in,out], Conv2D
[kh,kw,Cin,Cout], Conv2DTranspose [kh,kw,Cout,Cin]).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

_seed_state = {"seed": 0, "rng": np.random.default_rng(0)}


def set_seed(seed: int):
    """Counterpart of ``tf.random.set_seed`` (demo_mnist.py:96) for weight init and the step RNG."""
    _seed_state["seed"] = int(seed)
    _seed_state["rng"] = np.random.default_rng(int(seed))


def get_seed() -> int:
    return _seed_state["seed"]


def default_device():
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class Layer:
    kind = "layer"

    def __init__(self, input_shape=None, name=None, **kw):
        self._declared_input_shape = tuple(input_shape) if input_shape is not None else None
        self.name = name or self.__class__.__name__.lower()
        self.trainable = True
        self.vars = {}          # name -> torch view into the store (set by ParamStore.bind)

    # (name, shape, trainable, init) for an input shape (without batch)
    def var_specs(self, in_shape):
        return []

    def out_shape(self, in_shape):
        return tuple(in_shape)

    def count_params(self):
        return sum(int(v.numel()) for v in self.vars.values())


def _glorot(shape, fan_in, fan_out):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return lambda rng: rng.uniform(-lim, lim, shape).astype(np.float32)


_zeros = lambda shape: (lambda rng: np.zeros(shape, np.float32))
_ones = lambda shape: (lambda rng: np.ones(shape, np.float32))


class Dense(Layer):
    kind = "dense"

    def __init__(self, units, activation=None, use_bias=True, **kw):
        super().__init__(**kw)
        if activation not in (None, "linear"):
            raise NotImplementedError("Dense activation other than linear is not on the reference path")
        self.units, self.use_bias = int(units), bool(use_bias)

    def out_shape(self, in_shape):
        return (self.units,)

    def var_specs(self, in_shape):
        fan_in = int(in_shape[-1])
        specs = [("kernel", (fan_in, self.units), True, _glorot((fan_in, self.units), fan_in, self.units))]
        if self.use_bias:
            specs.append(("bias", (self.units,), True, _zeros((self.units,))))
        return specs


class _ConvBase(Layer):
    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", use_bias=True, activation=None, **kw):
        super().__init__(**kw)
        kh, kw_ = _pair(kernel_size)
        sh, sw = _pair(strides)
        if kh != kw_ or sh != sw:
            raise NotImplementedError("only square kernels / equal strides are on the reference path")
        if str(padding).lower() != "same":
            raise NotImplementedError("only padding='same' is on the reference path (demo_celeba.py:62-119)")
        if activation not in (None, "linear", "tanh"):
            raise NotImplementedError(f"activation {activation!r} is not on the reference path")
        self.filters, self.k, self.stride = int(filters), int(kh), int(sh)
        self.use_bias = bool(use_bias)
        self.activation = None if activation in (None, "linear") else activation


class Conv2D(_ConvBase):
    kind = "conv"

    def out_shape(self, s):
        return (-(-s[0] // self.stride), -(-s[1] // self.stride), self.filters)

    def var_specs(self, s):
        cin, k, f = int(s[2]), self.k, self.filters
        specs = [("kernel", (k, k, cin, f), True, _glorot((k, k, cin, f), k * k * cin, k * k * f))]
        if self.use_bias:
            specs.append(("bias", (f,), True, _zeros((f,))))
        return specs


class Conv2DTranspose(_ConvBase):
    kind = "convT"

    def out_shape(self, s):
        return (s[0] * self.stride, s[1] * self.stride, self.filters)

    def var_specs(self, s):
        cin, k, f = int(s[2]), self.k, self.filters
        specs = [("kernel", (k, k, f, cin), True, _glorot((k, k, f, cin), k * k * f, k * k * cin))]
        if self.use_bias:
            specs.append(("bias", (f,), True, _zeros((f,))))
        return specs


class BatchNormalization(Layer):
    kind = "bn"

    def __init__(self, momentum=0.99, epsilon=1e-3, **kw):
        super().__init__(**kw)
        self.momentum, self.epsilon = float(momentum), float(epsilon)

    def var_specs(self, s):
        c = int(s[-1])
        return [("gamma", (c,), True, _ones((c,))), ("beta", (c,), True, _zeros((c,))),
                ("moving_mean", (c,), False, _zeros((c,))), ("moving_variance", (c,), False, _ones((c,)))]


class LeakyReLU(Layer):
    kind = "lrelu"

    def __init__(self, alpha=0.3, **kw):
        super().__init__(**kw)
        self.alpha = float(alpha)


class Dropout(Layer):
    kind = "dropout"

    def __init__(self, rate, **kw):
        super().__init__(**kw)
        self.rate = float(rate)


class Reshape(Layer):
    kind = "reshape"

    def __init__(self, target_shape, **kw):
        super().__init__(**kw)
        self.target_shape = tuple(int(v) for v in target_shape)

    def out_shape(self, s):
        assert int(np.prod(s)) == int(np.prod(self.target_shape)), (s, self.target_shape)
        return self.target_shape


class Flatten(Layer):
    kind = "flatten"

    def out_shape(self, s):
        return (int(np.prod(s)),)


# ----------------------------------------------------------------------------------------------------
class ParamStore:
    """All variables of one top-level model in flat HBM buffers."""

    ALIGN = 4   # floats (16 bytes): every variable view is 16-byte aligned for float4 access

    def __init__(self, layers_with_shapes, device):
        self.device = torch.device(device)
        self.entries = []       # (layer, name, offset, shape, trainable)
        off_t = off_s = 0
        init_vals = []
        for layer, in_shape in layers_with_shapes:
            for name, shape, trainable, init in layer.var_specs(in_shape):
                n = int(np.prod(shape))
                old = layer.vars.get(name)
                val = old.detach().cpu().numpy().reshape(shape) if old is not None else init(_seed_state["rng"])
                if trainable:
                    self.entries.append((layer, name, off_t, tuple(shape), True))
                    off_t += -(-n // self.ALIGN) * self.ALIGN
                else:
                    self.entries.append((layer, name, off_s, tuple(shape), False))
                    off_s += -(-n // self.ALIGN) * self.ALIGN
                init_vals.append(val)
        self.n_train, self.n_state = off_t, off_s
        self.theta = torch.zeros(max(off_t, 1), dtype=torch.float32, device=self.device)
        self.state = torch.zeros(max(off_s, 1), dtype=torch.float32, device=self.device)
        self.grad = self.m = self.v = None
        self.step_count = 0
        for (layer, name, off, shape, trainable), val in zip(self.entries, init_vals):
            buf = self.theta if trainable else self.state
            n = int(np.prod(shape))
            view = buf[off:off + n].view(shape)
            view.copy_(torch.from_numpy(np.ascontiguousarray(val, dtype=np.float32)))
            layer.vars[name] = view
        # transposed copies of conv kernels ([taps][C][R] from [taps][R][C]) for the direction that needs them
        self._tr = {}           # id(layer) -> tensor
        self.tr_dirty = True

    def ensure_opt_state(self):
        if self.grad is None:
            self.grad = torch.zeros_like(self.theta)
            self.m = torch.zeros_like(self.theta)
            self.v = torch.zeros_like(self.theta)

    def view_like(self, buf, layer, name):
        cache = self.__dict__.setdefault("_view_cache", {})
        key = (buf.data_ptr(), id(layer), name)
        v = cache.get(key)
        if v is None:
            for (l, n, off, shape, trainable) in self.entries:
                if l is layer and n == name:
                    assert trainable
                    v = cache[key] = buf[off:off + int(np.prod(shape))].view(shape)
                    break
            else:
                raise KeyError(name)
        return v

    def train_range(self, *layers):
        """[lo, hi) of the flat trainable buffer covered by the variables of ``layers`` (None entries ignored)."""
        cache = self.__dict__.setdefault("_range_cache", {})
        key = tuple(id(x) for x in layers)
        r = cache.get(key)
        if r is None:
            lo, hi = None, None
            for (l, n, off, shape, trainable) in self.entries:
                if trainable and any(l is x for x in layers if x is not None):
                    end = off + -(-int(np.prod(shape)) // self.ALIGN) * self.ALIGN
                    lo = off if lo is None else min(lo, off)
                    hi = end if hi is None else max(hi, end)
            r = cache[key] = (0, 0) if lo is None else (lo, hi)
        return r

    def grad_of(self, layer, name):
        self.ensure_opt_state()
        return self.view_like(self.grad, layer, name)

    def transposed_kernel(self, layer):
        """[k*k][C][R] copy of a conv kernel stored [k*k][R][C]; refreshed after every optimiser step."""
        t = self._tr.get(id(layer))
        if t is None:
            raise KeyError("transposed kernels are laid out by refresh_transposed(); call Net.prepare_weights() first")
        return t

    def _plan_transposed(self, layers):
        """One flat buffer for every transposed copy + the device descriptor table of the batched transpose."""
        offs, desc, off, tile = {}, [], 0, 0
        for layer in layers:
            w = layer.vars["kernel"]
            k2, R, Cc = w.shape[0] * w.shape[1], w.shape[2], w.shape[3]
            src_off = (w.data_ptr() - self.theta.data_ptr()) // 4
            assert 0 <= src_off < self.theta.numel() and src_off % 4 == 0
            desc.append([src_off, off, k2, R, Cc, tile])
            offs[id(layer)] = (off, k2 * R * Cc)
            tile += k2 * (-(-R // 64)) * (-(-Cc // 64))
            off += -(-k2 * R * Cc // self.ALIGN) * self.ALIGN
        self._tr_flat = torch.empty(max(off, 4), dtype=torch.float32, device=self.device)
        self._tr = {k: self._tr_flat[o:o + n] for k, (o, n) in offs.items()}
        self._tr_desc = torch.tensor(desc, dtype=torch.int32, device=self.device).contiguous()
        self._tr_tiles = tile
        self._tr_layers = [id(l) for l in layers]

    def refresh_transposed(self, layers):
        from . import ops
        layers = list(layers)
        if not layers:
            self.tr_dirty = False
            return
        if getattr(self, "_tr_layers", None) != [id(l) for l in layers]:
            self._plan_transposed(layers)
        ops.transpose_last2_batched(self.theta, self._tr_flat, self._tr_desc, len(layers), self._tr_tiles)
        self.tr_dirty = False


class Sequential(Layer):
    """tf.keras.Sequential stand-in (demo_celeba.py:51,96)."""
    kind = "sequential"

    def __init__(self, layers: Optional[Sequence[Layer]] = None, name=None, **kw):
        super().__init__(name=name, **kw)
        self.layers: List[Layer] = []
        self._in_shape: Optional[Tuple[int, ...]] = None
        self._out_shape: Optional[Tuple[int, ...]] = None
        self._store: Optional[ParamStore] = None
        self._net = None
        self.optimizer = None
        for l in (layers or []):
            self.add(l)

    # ---- construction
    def add(self, layer: Layer):
        if self._store is not None:
            raise RuntimeError("cannot add layers to a built model")
        self.layers.append(layer)
        if self._in_shape is None:
            decl = layer._declared_input_shape
            if decl is None and isinstance(layer, Sequential):
                decl = layer._in_shape
            if decl is not None and len(self.layers) == 1:
                self._in_shape = tuple(decl)
                self._out_shape = tuple(decl)
        if self._in_shape is not None:
            self._out_shape = tuple(layer.out_shape(self._out_shape))

    def out_shape(self, s):
        for l in self.layers:
            s = l.out_shape(s)
        return tuple(s)

    @property
    def input_shape(self):
        return None if self._in_shape is None else (None,) + self._in_shape

    @property
    def output_shape(self):
        return None if self._out_shape is None else (None,) + self._out_shape

    def flat_layers(self):
        """[(layer, in_shape)] with nested Sequentials expanded."""
        out, s = [], self._in_shape
        if s is None:
            raise RuntimeError("model has no input shape: give the first layer input_shape=")
        for l in self.layers:
            if isinstance(l, Sequential):
                if l._in_shape is None:
                    l._in_shape, l._out_shape = tuple(s), l.out_shape(s)
                for ll, ss in l.flat_layers():
                    out.append((ll, ss))
            else:
                out.append((l, tuple(s)))
            s = l.out_shape(s)
        return out

    def build(self, device=None):
        """Allocates (or re-binds) the variables.  Wrapping built models into a new Sequential moves their
        variables into the wrapper's store and keeps the inner models bound to it."""
        if self._store is not None:
            return self
        flat = self.flat_layers()
        self._store = ParamStore(flat, device or default_device())
        for l in self.layers:
            if isinstance(l, Sequential):
                l._adopt(self._store)
        return self

    def _adopt(self, store):
        self._store, self._net = store, None
        for l in self.layers:
            if isinstance(l, Sequential):
                l._adopt(store)

    @property
    def store(self) -> ParamStore:
        self.build()
        return self._store

    def net(self):
        from .engine import Net
        if self._net is None:
            self.build()
            self._net = Net(self.flat_layers(), self._store)
        return self._net

    # ---- Keras surface
    def _own_layers(self):
        return [l for l, _ in self.flat_layers()]

    @property
    def trainable_variables(self):
        self.build()
        own = {id(l) for l in self._own_layers()}
        return [l.vars[n] for (l, n, _, _, tr) in self._store.entries if tr and id(l) in own]

    @property
    def variables(self):
        self.build()
        own = {id(l) for l in self._own_layers()}
        return [l.vars[n] for (l, n, _, _, tr) in self._store.entries if id(l) in own]

    def count_params(self):
        return sum(int(v.numel()) for v in self.variables)

    def get_weights(self):
        return [v.detach().cpu().numpy().copy() for v in self.variables]

    def set_weights(self, weights):
        vs = self.variables
        assert len(vs) == len(weights), (len(vs), len(weights))
        for v, w in zip(vs, weights):
            v.copy_(torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)).view(v.shape))
        self._store.tr_dirty = True

    def save_weights(self, filepath, overwrite=True, save_format=None):
        np.savez(filepath if str(filepath).endswith(".npz") else str(filepath) + ".npz",
                 **{f"v{i}": w for i, w in enumerate(self.get_weights())})

    def load_weights(self, filepath):
        p = filepath if str(filepath).endswith(".npz") else str(filepath) + ".npz"
        d = np.load(p)
        self.set_weights([d[f"v{i}"] for i in range(len(d.files))])

    def summary(self):
        print(f'Model: "{self.name}"')
        for l, s in self.flat_layers():
            print(f"  {l.__class__.__name__:<20} in={s!s:<18} out={l.out_shape(s)!s:<18} params={sum(int(np.prod(sh)) for _, sh, _, _ in l.var_specs(s)):,}")
        print(f"Total params: {self.count_params():,}")

    def __call__(self, x, training=False):
        """Forward pass on the HIP kernels; returns a fresh tensor [B, *output_shape]."""
        return self.net().predict(x, training=training)
