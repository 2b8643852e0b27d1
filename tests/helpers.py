"""Shared helpers of the GPU parity tests: oracle <-> product weight exchange, tolerances."""
import numpy as np
import torch

from oracle import models as OM

# Stated tolerances (SURVEY.md 8c proposal): fp32 HIP kernels vs the float64 oracle.
#   pointwise / blur: rtol 1e-5, atol 1e-6 ; conv family: rtol 2e-5*sqrt(K/800) of the output scale, floor 1e-5
#   one full step's updated weights: rtol 1e-4 (Adam's m/(sqrt(v)+eps) amplifies tiny-gradient noise -> atol 2e-5)
POINT_RTOL, POINT_ATOL = 1e-5, 1e-6


def conv_tol(K, scale):
    return max(2e-5 * np.sqrt(max(K, 1) / 800.0), 1e-5) * max(scale, 1e-6) * 4


def oracle_weight_list(params):
    """Oracle per-layer dicts -> flat list in Keras variable order."""
    out = []
    for p in params:
        for k in ("kernel", "bias", "gamma", "beta", "moving_mean", "moving_var"):
            if k in p:
                out.append(np.asarray(p[k], dtype=np.float32))
    return out


def load_oracle_weights(model, params):
    model.build()
    model.set_weights(oracle_weight_list(params))


def product_grads(model):
    """Flat list of gradient arrays in trainable-variable order."""
    st = model.store
    own = {id(l) for l in model._own_layers()}
    return [st.view_like(st.grad, l, n).detach().cpu().numpy().copy() for (l, n, _, _, tr) in st.entries if tr and id(l) in own]


def oracle_grad_list(grads):
    out = []
    for g in grads:
        for k in ("kernel", "bias", "gamma", "beta"):
            if k in g:
                out.append(np.asarray(g[k]))
    return out


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype)


def product_slots(model, which):
    """Adam slot ('m' or 'v') arrays in trainable-variable order."""
    st = model.store
    own = {id(l) for l in model._own_layers()}
    buf = getattr(st, which)
    return [st.view_like(buf, l, n).detach().cpu().numpy().copy() for (l, n, _, _, tr) in st.entries if tr and id(l) in own]


def sync_oracle_from_product(st, gan):
    """Overwrites the oracle state's weights, BN statistics and Adam slots with the PRODUCT's current values (float64
    copies of its float32 buffers), so that the next oracle step starts from exactly the state the product is in."""
    for model, key in ((gan.generator, "g"), (gan.discriminator, "d")):
        it = iter(model.get_weights())
        for p in st[key]:
            for k in ("kernel", "bias", "gamma", "beta", "moving_mean", "moving_var"):
                if k in p:
                    p[k] = next(it).astype(np.float64).reshape(p[k].shape)
        for slot in ("m", "v"):
            it = iter(product_slots(model, slot))
            for p in st[f"{key}_{slot}"]:
                for k in ("kernel", "bias", "gamma", "beta"):
                    if k in p:
                        p[k] = next(it).astype(np.float64).reshape(p[k].shape)
    return st
