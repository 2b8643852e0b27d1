"""Shared helpers of the GPU parity tests: oracle <-> product weight exchange, tolerances."""
import numpy as np
import torch

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


def rel_l2(a, b):
    """||a - b||_2 / ||b||_2 in float64: the well-conditioned companion of the elementwise bounds."""
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def cosine(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))


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


# Well-conditioned companions of the elementwise step bounds: per-tensor relative L2 error and cosine against the
# float64 oracle.  A small systematic kernel error (a wrong tap weight, a missing term) moves these by orders of magnitude,
# while the float32 cancellation noise of BatchNorm's backward, which forced the elementwise bounds up, does not.
#
# Base bounds: relative L2 <= 1e-3 (generator) / 2e-4 (critic), cosine >= 1 - 1e-6.  Where a gradient is itself a small
# residue of large cancelling terms no float32 evaluation meets a fixed bound -- measured on the CPU, the ORACLE run in float32
# against its own float64 run at celeba64 / batch 64: generator 8e-4 ... 7e-3 (1 - cos up to 2e-5), critic <= 1.5e-5, the Dense
# bias of the critic (sum of +-B/gbs + 1e-4 sign terms, a 1e-4-sized residue) 5.5e-4.  So a test may pass that float32 run as
# the YARDSTICK of the precision class: the bound of a variable is then max(base, 3 x the float32 oracle's own deviation).
GRAD_L2 = {"g": 1e-3, "d": 2e-4}
GRAD_COS = 1e-6
YARDSTICK = 3.0


def to_float32_state(st):
    """Copy of an oracle state with every array in float32 (weights, BN statistics, Adam slots)."""
    import copy
    st32 = copy.deepcopy(st)
    for key in ("g", "d", "g_m", "g_v", "d_m", "d_v"):
        st32[key] = [{k: np.asarray(v).astype(np.float32) for k, v in p.items()} for p in st32[key]]
    return st32


def to_float32_randomness(rnd):
    return {k: (v.astype(np.float32) if isinstance(v, np.ndarray) else v) for k, v in rnd.items()}


def check_grad_quality(prod, ora, key, label, ora32=None, l2_bound=None, cos_bound=GRAD_COS):
    """prod / ora: gradient lists of one network ('g' / 'd'), HIP path and float64 oracle; ora32: the float32 oracle's (the
    yardstick, optional).  Returns {index: (rel L2, 1 - cosine)}; raises with the whole table when a variable exceeds
    max(base bound, 3 x yardstick).  A variable whose float64 gradient is exactly zero must be (nearly) zero in the product."""
    l2_bound = GRAD_L2[key] if l2_bound is None else l2_bound
    table, bad, cells = {}, [], []
    for i, (a, b) in enumerate(zip(prod, ora)):
        b = np.asarray(b, np.float64).reshape(a.shape)
        if not np.any(b):
            ok = float(np.abs(a).max()) <= 1e-6
            table[i] = (float(np.abs(a).max()), 0.0)
            cells.append(f"{key}{i:02d} zero-gradient |a|max {table[i][0]:.1e}")
            bad += [] if ok else [i]
            continue
        l2, c = rel_l2(a, b), 1.0 - cosine(a, b)
        lb, cb, y = l2_bound, cos_bound, ""
        if ora32 is not None:
            b32 = np.asarray(ora32[i], np.float64).reshape(a.shape)
            l32, c32 = rel_l2(b32, b), 1.0 - cosine(b32, b)
            lb, cb = max(lb, YARDSTICK * l32), max(cb, YARDSTICK * c32)
            y = f" [f32 oracle {l32:.1e}/{c32:.0e}]"
        table[i] = (l2, c)
        cells.append(f"{key}{i:02d} {l2:.1e}/{c:.0e}{y}")
        if not (l2 <= lb and c <= cb):
            bad.append(i)
    line = ", ".join(cells)
    print(f"[grad quality] {label}: rel-L2 / (1-cos) per variable: {line}")
    assert not bad, f"{label}: variables {bad} exceed max(rel-L2 {l2_bound:g}, 1-cos {cos_bound:g}; {YARDSTICK:g} x float32 oracle): {line}"
    return table


def arm_branch_capture(gan):
    """Call BEFORE train_on_batch: the merged second-order pass overwrites the x-hat rows of the critic's activations in place
    (engine.Net.gp_second_order_merged); with this list armed it first copies their signs out (instrumentation only -- the
    arithmetic path is the default one)."""
    gan.discriminator.net().capture_branches = []


def product_lrelu_branches(gan, B):
    """LeakyReLU branch decisions (1 / alpha per unit) the product took in its LAST train_on_batch, read back from the
    activations its passes left in their contexts (merged critic pass "fr3": rows [0, B) fakes, [B, 2B) reals, [2B, 3B) x-hat
    -- those from the signs captured by arm_branch_capture when the merged second-order pass has overwritten them;
    G-step: generator "g", critic "hat").  Shaped for oracle.step's ``force`` argument.  A unit that Dropout zeroed reads as
    alpha here and is multiplied by its keep mask (0) in the oracle's backward anyway."""
    G, D = gan.generator.net(), gan.discriminator.net()

    def masks(net, ctx, lo=None, hi=None):
        out = []
        for i, st in enumerate(net.stages):
            if st.act == "lrelu":
                a = ctx.a[i] if lo is None else ctx.a[i][lo:hi]
                out.append(np.where(a.detach().cpu().numpy() > 0, 1.0, float(st.alpha)))
        return out
    c3 = D.context(3 * B, "fr3", drop_rows=2 * B)
    if D.capture_branches:
        alphas = [float(st.alpha) for st in D.stages if st.act == "lrelu"]
        hat = [np.where(s, 1.0, a) for s, a in zip(D.capture_branches, alphas)]
        D.capture_branches = None
    else:
        assert not getattr(gan, "merge_gp_filter_gradients", False) or not gan.uses_gradient_penalty, "arm_branch_capture(gan) first"
        hat = masks(D, c3, 2 * B, 3 * B)
    return {"fake": masks(D, c3, 0, B), "real": masks(D, c3, B, 2 * B), "hat": hat,
            "g": masks(G, G.context(B, "g")), "d_gstep": masks(D, D.context(B, "hat"))}
