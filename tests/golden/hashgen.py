"""Counter-based deterministic numbers for fixtures that are too large to store (the 2.5 M weights of the MNIST stacks).

Pure integer arithmetic on uint64 (splitmix64 finaliser of ``seed * 2^32 + index``), so the values do not depend on the numpy
version's bit generators.  A fixture that uses it stores the seed AND per-variable checksums of the generated values; the
loader verifies the checksums before use, so the inputs are pinned by the file, not by this code."""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def hashed_uniform(seed, n, lo=-1.0, hi=1.0):
    """n float64 values in [lo, hi), element i a function of (seed, i) only."""
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(1 << 32) + np.uint64(1)) * _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return lo + (hi - lo) * u


def hashed_indices(seed, n, count):
    """``count`` distinct sorted indices below n (all of them when n <= count)."""
    if n <= count:
        return np.arange(n, dtype=np.int64)
    u = hashed_uniform(seed, 4 * count, 0.0, 1.0)
    idx = np.unique((u * n).astype(np.int64))
    return idx[:: max(1, len(idx) // count)][:count]


WKEYS = ("kernel", "bias", "gamma", "beta", "moving_mean", "moving_var")     # Keras variable order inside one layer


def hashed_variable(kind, shape, seed):
    """One variable: kernels Glorot-uniform-scaled, gamma / moving_var 1 + 0.2 |u|, everything else 0.1 u; rounded to
    float32 (what the product stores), returned as float64."""
    shape = tuple(int(d) for d in shape)
    u = hashed_uniform(seed, int(np.prod(shape))).reshape(shape)
    if kind == "kernel":
        lim = np.sqrt(6.0 / (shape[0] + shape[1])) if len(shape) == 2 else np.sqrt(6.0 / (shape[0] * shape[1] * (shape[2] + shape[3])))
        a = u * lim
    elif kind in ("gamma", "moving_var"):
        a = 1.0 + 0.2 * np.abs(u)
    else:
        a = 0.1 * u
    return a.astype(np.float32).astype(np.float64)


def hashed_params(spec_params, seed):
    """Replaces every array of an oracle parameter list (list of per-layer dicts) by hashed values of the same shape;
    variable number i (Keras order) uses stream ``seed + i``.  Returns (params, kinds)."""
    out, kinds = [], []
    for p in spec_params:
        q = {}
        for k in WKEYS:
            if k in p:
                q[k] = hashed_variable(k, p[k].shape, seed + len(kinds)).astype(p[k].dtype)
                kinds.append(k)
        out.append(q)
    return out, kinds


def hashed_weight_list(kinds, shapes, seed):
    """The same values as ``hashed_params`` from the stored variable kinds and shapes alone (no oracle needed)."""
    return [hashed_variable(k, s, seed + i) for i, (k, s) in enumerate(zip(kinds, shapes))]


def checksum(a):
    a = np.asarray(a, dtype=np.float64).ravel()
    w = 1.0 + (np.arange(a.size) % 7)
    return np.array([a.sum(), (a * a).sum(), (a * w).sum()])
