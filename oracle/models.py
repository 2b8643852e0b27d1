"""Neutral layer-stack specs + a small numpy interpreter (forward / backward / linearised forward).

Oracle = test infrastructure (see ``oracle/__init__.py``); parity unpinned.

Specs follow the reference demos:
  * ``mnist``      demo_mnist.py:48-86
  * ``celeba128``  demo_celeba.py:51-124 (verbatim)
  * ``celeba64``   build-side definition (SURVEY.md 8a "Architecture note"): the 128 stack minus
                   its outermost stage on each side.
A spec is a list of dicts; parameters are a parallel list of dicts (Keras variable names).
"""
from __future__ import annotations

import numpy as np
from . import np_ops as O


def _g_stack(base_hw, dense_ch, convt, last):
    spec = [dict(type="dense", units=base_hw * base_hw * dense_ch, use_bias=False),
            dict(type="bn"), dict(type="lrelu"),
            dict(type="reshape", shape=(base_hw, base_hw, dense_ch))]
    for filters, stride, act in convt:
        spec.append(dict(type="convT", filters=filters, k=5, stride=stride, use_bias=False, activation=act))
        if act is None:
            spec += [dict(type="bn"), dict(type="lrelu")]
    if last is not None:
        spec.append(dict(type="conv", filters=last, k=5, stride=1, use_bias=False, activation="tanh"))
    return spec


def _d_stack(chs):
    spec = []
    for c in chs:
        spec += [dict(type="conv", filters=c, k=5, stride=2, use_bias=True, activation=None),
                 dict(type="lrelu"), dict(type="dropout", rate=0.3)]
    spec += [dict(type="flatten"), dict(type="dense", units=1, use_bias=True)]
    return spec


def generator_spec(name):
    if name == "mnist":      # demo_mnist.py:48-71
        return _g_stack(7, 256, [(128, 1, None), (64, 2, None), (1, 2, "tanh")], None)
    if name == "celeba128":  # demo_celeba.py:51-93
        return _g_stack(4, 512, [(512, 1, None), (256, 2, None), (128, 2, None), (64, 2, None),
                                 (32, 2, None), (16, 2, None)], 3)
    if name == "celeba64":
        return _g_stack(4, 512, [(512, 1, None), (256, 2, None), (128, 2, None), (64, 2, None),
                                 (32, 2, None)], 3)
    if name == "tiny":       # test-only: 8x8x3 images, same layer types
        return _g_stack(2, 32, [(32, 1, None), (16, 2, None), (16, 2, None)], 3)
    if name == "tiny_mnist":  # test-only: ConvT with fused tanh and 1 channel, odd sizes
        return _g_stack(3, 8, [(8, 1, None), (4, 2, None), (1, 2, "tanh")], None)
    raise KeyError(name)


def discriminator_spec(name):
    if name == "mnist":      # demo_mnist.py:74-86
        return _d_stack([64, 128])
    if name == "celeba128":  # demo_celeba.py:96-124
        return _d_stack([16, 32, 64, 128, 256, 512])
    if name == "celeba64":
        return _d_stack([32, 64, 128, 256, 512])
    if name == "tiny":
        return _d_stack([16, 32])
    if name == "tiny_mnist":
        return _d_stack([4, 8])
    raise KeyError(name)


def image_shape(name):
    return {"mnist": (28, 28, 1), "celeba128": (128, 128, 3), "celeba64": (64, 64, 3),
            "tiny": (8, 8, 3), "tiny_mnist": (12, 12, 1)}[name]


LATENT = {"mnist": 100, "celeba128": 100, "celeba64": 100, "tiny": 10, "tiny_mnist": 6}


# ----------------------------------------------------------------------------
# shapes + init ([TF] glorot_uniform kernels, zero bias, BN gamma=1 beta=0 mean=0 var=1)
# ----------------------------------------------------------------------------
def infer_shapes(spec, in_shape):
    """Returns the per-layer output shapes (without batch)."""
    shapes, s = [], tuple(in_shape)
    for L in spec:
        t = L["type"]
        if t == "dense":
            s = (L["units"],)
        elif t == "reshape":
            s = tuple(L["shape"])
        elif t == "flatten":
            s = (int(np.prod(s)),)
        elif t == "conv":
            s = (-(-s[0] // L["stride"]), -(-s[1] // L["stride"]), L["filters"])
        elif t == "convT":
            s = (s[0] * L["stride"], s[1] * L["stride"], L["filters"])
        shapes.append(s)
    return shapes


def init_params(spec, in_shape, rng, dtype=np.float32):
    params, s = [], tuple(in_shape)
    shapes = infer_shapes(spec, in_shape)
    for L, so in zip(spec, shapes):
        t, p = L["type"], {}
        if t == "dense":
            fan_in, fan_out = s[0], L["units"]
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            p["kernel"] = rng.uniform(-lim, lim, (fan_in, fan_out)).astype(dtype)
            if L["use_bias"]:
                p["bias"] = np.zeros(fan_out, dtype)
        elif t in ("conv", "convT"):
            k, cin, f = L["k"], s[2], L["filters"]
            shp = (k, k, cin, f) if t == "conv" else (k, k, f, cin)
            lim = np.sqrt(6.0 / (k * k * (cin + f)))
            p["kernel"] = rng.uniform(-lim, lim, shp).astype(dtype)
            if L["use_bias"]:
                p["bias"] = np.zeros(f, dtype)
        elif t == "bn":
            c = s[-1]
            p = dict(gamma=np.ones(c, dtype), beta=np.zeros(c, dtype),
                     moving_mean=np.zeros(c, dtype), moving_var=np.ones(c, dtype))
        params.append(p)
        s = so
    return params


TRAINABLE = ("kernel", "bias", "gamma", "beta")


# ----------------------------------------------------------------------------
# interpreter
# ----------------------------------------------------------------------------
def forward(spec, params, x, training, masks=None, update_bn=None, lrelu_masks=None):
    """Runs the stack.  ``masks``: list of keep-masks, one per dropout layer, used when training.
    ``update_bn``: dict collecting new moving stats (layer index -> (mean, var)) when training.
    ``lrelu_masks`` (test aid): one array of LeakyReLU derivatives (1 / alpha) per LeakyReLU layer, used INSTEAD of the signs this
    pass computes -- the gradient of the network is discontinuous where a pre-activation crosses zero, and a float32
    implementation within 1e-7 of the kink may sit on the other branch; with the branches of the implementation under test
    forced, the comparison is between the same piecewise-linear function.  Forward VALUES still follow this pass's own signs.
    Returns (out, cache)."""
    cache, mi, li = [], 0, 0
    for i, (L, p) in enumerate(zip(spec, params)):
        t = L["type"]
        c = {"x": x}
        if t == "dense":
            x = O.dense_fwd(x, p["kernel"], p.get("bias"))
        elif t == "reshape":
            x = x.reshape((x.shape[0],) + tuple(L["shape"]))
        elif t == "flatten":
            x = x.reshape(x.shape[0], -1)
        elif t in ("conv", "convT"):
            if t == "conv":
                x = O.conv2d_fwd(x, p["kernel"], L["stride"])
            else:
                x = O.conv2d_transpose_fwd(x, p["kernel"], L["stride"])
            if L["use_bias"]:
                x = x + p["bias"]
            if L.get("activation") == "tanh":
                x = np.tanh(x)
                c["y"] = x
        elif t == "bn":
            if training:
                x, bc, nm, nv = O.bn_train_fwd(x, p["gamma"], p["beta"], p["moving_mean"], p["moving_var"])
                c["bn"] = bc
                if update_bn is not None:
                    update_bn[i] = (nm, nv)
            else:
                x = O.bn_infer_fwd(x, p["gamma"], p["beta"], p["moving_mean"], p["moving_var"])
        elif t == "lrelu":
            c["m"] = O.lrelu_mask(x) if lrelu_masks is None else np.asarray(lrelu_masks[li], dtype=x.dtype).reshape(x.shape)
            li += 1
            x = O.lrelu_fwd(x)
        elif t == "dropout":
            if training:
                c["keep"] = masks[mi]
                x = O.dropout_fwd(x, masks[mi], L["rate"])
            mi += 1
        else:
            raise KeyError(t)
        cache.append(c)
    return x, cache


def backward(spec, params, cache, dout, need_dx=True, need_dw=True, training=True, keep_dz=None):
    """Explicit reverse pass.  Returns (grads list of dicts, dx).  ``keep_dz``: dict that receives the
    gradient w.r.t. each conv layer's pre-activation (zeta_i of SURVEY.md 8a, GP derivation)."""
    grads = [dict() for _ in spec]
    d = dout
    for i in range(len(spec) - 1, -1, -1):
        L, p, c = spec[i], params[i], cache[i]
        t = L["type"]
        first = (i == 0)
        want_dx = need_dx or not first
        if t == "dense":
            dx, dw, db = O.dense_bwd(c["x"], p["kernel"], d, need_dx=want_dx)
            if need_dw:
                grads[i]["kernel"] = dw
                if L["use_bias"]:
                    grads[i]["bias"] = db
            d = dx
        elif t in ("reshape", "flatten"):
            d = d.reshape(c["x"].shape)
        elif t in ("conv", "convT"):
            if L.get("activation") == "tanh":
                d = d * (1 - c["y"] ** 2)
            if keep_dz is not None:
                keep_dz[i] = d
            x = c["x"]
            if need_dw:
                if t == "conv":
                    grads[i]["kernel"] = O.conv2d_bwd_filter(x, d, L["stride"], L["k"])
                else:
                    grads[i]["kernel"] = O.conv2d_transpose_bwd_filter(x, d, L["stride"], L["k"])
                if L["use_bias"]:
                    grads[i]["bias"] = d.sum((0, 1, 2))
            if want_dx:
                if t == "conv":
                    d = O.conv2d_bwd_data(d, p["kernel"], L["stride"], x.shape[1:3])
                else:
                    d = O.conv2d_transpose_bwd_data(d, p["kernel"], L["stride"])
            else:
                d = None
        elif t == "bn":
            if training:
                d, dg, db = O.bn_train_bwd(d, p["gamma"], c["bn"])
                if need_dw:
                    grads[i]["gamma"], grads[i]["beta"] = dg, db
            else:
                inv = 1.0 / np.sqrt(p["moving_var"] + d.dtype.type(O.BN_EPS))
                if need_dw:
                    xh = (c["x"] - p["moving_mean"]) * inv
                    ax = tuple(range(d.ndim - 1))
                    grads[i]["gamma"], grads[i]["beta"] = (d * xh).sum(ax), d.sum(ax)
                d = d * p["gamma"] * inv
        elif t == "lrelu":
            d = d * c["m"]
        elif t == "dropout":
            if "keep" in c:
                d = O.dropout_fwd(d, c["keep"], L["rate"])
    return grads, d


def linearised_forward(spec, params, cache, dz, v):
    """GP second order ("GP second-order derivation" step 2), critic stacks only:
    pushes ``v`` (= delta-bar_0) through the critic linearised at the x-hat pass (frozen LeakyReLU
    masks, dropout inactive, no biases) and collects dGP/dW_i = wgrad(x := delta-bar_{i-1}, dy := zeta_i).
    The last Dense(->1) gets sum_s flat(delta-bar_L)_s."""
    grads = [dict() for _ in spec]
    for i, (L, p, c) in enumerate(zip(spec, params, cache)):
        t = L["type"]
        if t == "conv":
            grads[i]["kernel"] = O.conv2d_bwd_filter(v, dz[i], L["stride"], L["k"])
            if L["use_bias"]:
                grads[i]["bias"] = np.zeros_like(p["bias"])
            v = O.conv2d_fwd(v, p["kernel"], L["stride"])
        elif t == "lrelu":
            v = v * c["m"]
        elif t == "dropout":
            pass
        elif t == "flatten":
            v = v.reshape(v.shape[0], -1)
        elif t == "dense":
            assert L["units"] == 1
            grads[i]["kernel"] = v.sum(0)[:, None]
            if L["use_bias"]:
                grads[i]["bias"] = np.zeros_like(p["bias"])
        else:
            raise KeyError(t)
    return grads
