"""Per-layer gradient error of one product step against the fp64 oracle (max |err| / max |ref| per variable).
Usage: python tests/step_error.py [arch] [B] [std]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from oracle import step as S
import test_step_gpu as T

arch = sys.argv[1] if len(sys.argv) > 1 else "celeba64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
std = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
gan, st, reals, rng = T._make(arch, B, std, gbs=B + 1)
rnd = S.draw_randomness(arch, B, rng, np.float64)
hp = dict(S.DEFAULT_HP, global_batch_size=B + 1)
dg, met, fakes = S.discriminator_grads(st, reals, rnd, hp)
gan.discriminator.optimizer.learning_rate = 0.0
gan.generator.optimizer.learning_rate = 0.0
gan.train_on_batch(reals.astype(np.float32), randomness=rnd)
for name, prod, ora in (("D", T.product_grads(gan.discriminator), T.oracle_grad_list(dg)),
                        ("G", T.product_grads(gan.generator), T.oracle_grad_list(S.generator_grads(st, rnd, hp, B)[0]))):
    for i, (a, b) in enumerate(zip(prod, ora)):
        b = b.reshape(a.shape)
        print(f"{name}{i:02d} shape {str(a.shape):24s} max|ref| {np.abs(b).max():.3e}  max|err|/max|ref| {np.abs(a - b).max() / max(np.abs(b).max(), 1e-30):.3e}")
