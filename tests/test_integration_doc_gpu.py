"""INTEGRATION.md is executable: the ctypes stubs a maintainer of the reference would paste (section B) are extracted from the
document and run here -- the blur drop-in against the float64 oracle, the step-program example against the eager calls."""
import math
import os
import re

import numpy as np
import pytest
import torch

from oracle import np_ops as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```python\n(.*?)```", text, flags=re.S)


def _namespace(monkeypatch):
    monkeypatch.chdir(ROOT)                          # the stub opens the library by its in-tree relative path
    blocks = _blocks()
    blur = next(b for b in blocks if "def blur_images" in b)
    prog = next(b for b in blocks if "def record_toy_step" in b)
    ns = {}
    exec(compile(blur, "INTEGRATION.md:blur", "exec"), ns)
    exec(compile(prog, "INTEGRATION.md:program", "exec"), ns)
    return ns


@pytest.mark.parametrize("shape,scale", [((3, 64, 64, 3), 5.0), ((2, 28, 28, 1), 1.0), ((1, 128, 128, 3), 23.5)])
def test_blur_stub_of_the_document(monkeypatch, shape, scale):
    ns = _namespace(monkeypatch)
    x = np.random.default_rng(0).uniform(-1, 1, size=shape).astype(np.float32)
    y = ns["blur_images"](torch.from_numpy(x).cuda(), scale).cpu().numpy()
    np.testing.assert_allclose(y, O.blur_images(x.astype(np.float64), scale), rtol=0, atol=5e-6)


def test_step_program_stub_of_the_document(monkeypatch):
    from blurred_gan_amd import ops
    ns = _namespace(monkeypatch)
    n, seed = 4096, 99
    stream = torch.cuda.current_stream().cuda_stream
    theta, m, v, g = (torch.zeros(n, device="cuda") for _ in range(4))
    theta.fill_(0.5)
    prog, f64, u64 = ns["record_toy_step"](theta, m, v, g, seed, stream)      # the recording run IS step 1 (offset 0, lr_t 1e-3)
    want_t, want_m, want_v = (torch.zeros(n, device="cuda") for _ in range(3))
    want_t.fill_(0.5)
    wg = torch.zeros(n, device="cuda")
    ops.uniform(wg, seed, 0)
    ops.adam(want_t, want_m, want_v, wg, 1e-3)
    assert torch.equal(theta, want_t) and torch.equal(g, wg)
    for step in (2, 3):
        off = (step - 1) * (n // 4)
        lr_t = 1e-3 * math.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
        ns["replay_toy_step"](prog, f64, u64, off, lr_t, stream)
        ops.uniform(wg, seed, off)
        ops.adam(want_t, want_m, want_v, wg, lr_t)
        assert torch.equal(g, wg) and torch.equal(theta, want_t) and torch.equal(m, want_m) and torch.equal(v, want_v), step
    ns["_lib"].bg_program_destroy(prog)
