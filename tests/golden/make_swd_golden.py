"""Generates tests/golden/swd_golden.npz by running the REFERENCE module /root/reference/sliced_wasserstein.py
(numpy + scipy only; the one hot-path-adjacent reference module importable in the build container)
on small seeded inputs.  Only inputs and outputs are stored -- no reference source.  Run from the repo root:
    python tests/golden/make_swd_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
import sliced_wasserstein as ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.RandomState(1234)
    batch = rng.uniform(0, 255, size=(3, 3, 32, 32)).astype(np.float64)          # float64: the reference copies it
    pyr = ref.generate_laplacian_pyramid(batch, 2)
    up = ref.pyr_up(rng.uniform(-1, 1, size=(2, 3, 5, 6)).astype(np.float32).copy())
    rng2 = np.random.RandomState(77)
    small = rng2.uniform(-1, 1, size=(2, 3, 5, 6)).astype(np.float32)
    out = dict(batch=batch, pyr0=pyr[0], pyr1=pyr[1], small=small, small_up=ref.pyr_up(small.copy()), small_down=ref.pyr_down(
        rng2.uniform(-1, 1, size=(2, 3, 8, 10)).astype(np.float32)))
    rng3 = np.random.RandomState(77)
    rng3.uniform(-1, 1, size=(2, 3, 5, 6))
    out["down_in"] = rng3.uniform(-1, 1, size=(2, 3, 8, 10)).astype(np.float32)
    # descriptors with the global generator seeded
    level = rng.uniform(-1, 1, size=(4, 3, 16, 16)).astype(np.float32)
    np.random.seed(4321)
    desc = ref.get_descriptors_for_minibatch(level, 7, 5)
    out.update(level=level, desc=desc, desc_final=ref.finalize_descriptors(desc.copy()))
    A = rng.normal(size=(64, 147)).astype(np.float32)
    B = (rng.normal(size=(64, 147)) * 1.3 + 0.2).astype(np.float32)
    np.random.seed(999)
    out.update(A=A, B=B, swd=np.float64(ref.sliced_wasserstein(A, B, 3, 16)))
    # full API run
    reals = rng.uniform(0, 255, size=(4, 3, 32, 32)).astype(np.float64)
    fakes = rng.uniform(0, 255, size=(4, 3, 32, 32)).astype(np.float64) * 0.8
    np.random.seed(2024)
    api = ref.API((4, 32, 32, 3))
    api.begin("reals"); api.feed("reals", reals); api.end("reals")
    api.begin("fakes"); api.feed("fakes", fakes); res = api.end("fakes")
    out.update(api_reals=reals, api_fakes=fakes, api_result=np.array(res, np.float64))
    np.savez_compressed(os.path.join(HERE, "swd_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "swd_golden.npz"), {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
