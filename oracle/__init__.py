"""CPU oracle for the blurred-GAN WGAN-GP training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.
The product path (``blurred-gan_amd/``) never imports this package and fails
loudly when its HIP library is missing.

PARITY UNPINNED.  The reference (lebrice/blurred-GAN) has no tests, golden
vectors or fixtures for this path, and its arithmetic
lives in TensorFlow 2.0 (``tensorflow-gpu==2.0.0rc1``, reference
``environment.yml:95``), which is not installed here, so the reference cannot
be run to generate vectors.  This oracle is therefore a restatement of the
reference's algorithm from its source text plus TensorFlow's documented op
semantics, pinned only by (a) analytic known-answer values, (b) an independent
torch-CPU autograd implementation of the same maths (``oracle/torch_ref.py``)
and (c) the reference's own shape asserts (``demo_celeba.py:60-93``,
``demo_mnist.py:58-71``).

Modules
-------
np_ops     explicit numpy forward/backward formulas for every op on the path
models     the DCGAN layer stacks of the reference demos as neutral specs
step       one full ``train_on_batch`` (D-step with GP second order, G-step, Adam)
torch_ref  the same step written with torch-CPU ops + autograd (cross-check and
           the ``cpu_baseline`` timed by bench.py)
"""
