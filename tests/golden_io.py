"""Readers of the frozen hot-path fixtures (tests/golden/hotpath_*.npz, written by tests/golden/make_hotpath_golden.py).
Imports neither the oracle nor the product: both are checked AGAINST these files."""
import os
import sys

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)
import hashgen as Hg  # noqa: E402

SAMPLE = 4096


def load_ops():
    return np.load(os.path.join(GOLDEN, "hotpath_ops.npz"))


class StepFixture:
    def __init__(self, arch):
        self.z = np.load(os.path.join(GOLDEN, f"hotpath_step_{arch}.npz"))
        z = self.z
        self.arch, self.B, self.sigma, self.steps = str(z["arch"]), int(z["B"]), float(z["sigma"]), int(z["steps"])
        self.hp = dict(zip([str(k) for k in z["hp_names"]], [float(v) for v in z["hp_values"]]))
        self.hp["d_steps_per_g_step"] = int(self.hp["d_steps_per_g_step"])
        self.hp["global_batch_size"] = int(self.hp["global_batch_size"])
        self.full = "g_w00" in z.files

    def kinds(self, key):
        return [str(k) for k in self.z[f"{key}_kinds"]]

    def weights(self, key):
        """Initial variables of network ``key`` ('g' / 'd') in Keras order, float64 holding float32 values.  Stored in the
        file, or regenerated from the stored seed -- either way verified against the stored checksums."""
        z, n = self.z, int(self.z[f"{key}_nvars"])
        shapes = [tuple(int(d) for d in z[f"{key}_w{i:02d}_shape"]) for i in range(n)]
        if self.full:
            ws = [z[f"{key}_w{i:02d}"].astype(np.float64) for i in range(n)]
        else:
            ws = Hg.hashed_weight_list(self.kinds(key), shapes, int(z[f"{key}_hash_seed"]))
        for i, w in enumerate(ws):
            assert w.shape == shapes[i]
            np.testing.assert_allclose(Hg.checksum(w), z[f"{key}_w{i:02d}_check"], rtol=1e-12, atol=1e-12,
                                       err_msg=f"fixture input {key} variable {i} does not reproduce")
        return ws

    def trainable(self, key):
        return [k in ("kernel", "bias", "gamma", "beta") for k in self.kinds(key)]

    def reals(self, it):
        return self.z[f"s{it}_reals"]

    def randomness(self, it, dtype=np.float64):
        z = self.z
        rnd = {k: z[f"s{it}_{k}"].astype(dtype) for k in ("z_d", "z_g", "alpha")}
        for k in ("mask_fake", "mask_real"):
            ms, j = [], 0
            while f"s{it}_{k}{j}" in z.files:
                shape = tuple(int(d) for d in z[f"s{it}_{k}{j}_shape"])
                ms.append(np.unpackbits(z[f"s{it}_{k}{j}"])[: int(np.prod(shape))].reshape(shape).astype(np.uint8))
                j += 1
            rnd[k] = ms
        return rnd

    def metrics(self, it):
        return dict(zip([str(k) for k in self.z[f"s{it}_metric_names"]], [float(v) for v in self.z[f"s{it}_metrics"]]))

    def sample_index(self, i, size):
        return np.arange(size) if self.full else Hg.hashed_indices(77 + i, size, SAMPLE)

    def grads(self, it, key):
        """[(sampled expected gradient, L2 norm of the whole gradient, max |g|)] per trainable variable."""
        out, i = [], 0
        while f"s{it}_{key}_grad{i:02d}" in self.z.files:
            out.append((self.z[f"s{it}_{key}_grad{i:02d}"], float(self.z[f"s{it}_{key}_grad{i:02d}_norm"]),
                        float(self.z[f"s{it}_{key}_grad{i:02d}_max"])))
            i += 1
        return out

    def after(self, it, key):
        """[(sampled expected variable after step it, checksum of the whole variable)] per variable (Keras order)."""
        n = int(self.z[f"{key}_nvars"])
        return [(self.z[f"s{it}_{key}_after{i:02d}"], self.z[f"s{it}_{key}_after{i:02d}_check"]) for i in range(n)]


def rel_l2(a, b):
    """||a - b||_2 / ||b||_2 in float64."""
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def cosine(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))
