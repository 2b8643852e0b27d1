"""Per-variable gradient error of the product against the fp64 oracle, with the float32 oracle beside it as the yardstick.
Runs `steps` consecutive steps (real learning rate); before every step after the first the oracle is re-synchronised to the
product's state, so each step's gradients are compared from IDENTICAL inputs.  Gradients are read back from Adam's first moment:
g = (m_new - 0.9 m_old) / 0.1.
Usage: python tests/step_error.py [arch] [B] [std] [steps] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from oracle import step as S
import test_step_gpu as T
from helpers import (product_slots, oracle_grad_list, rel_l2, cosine, sync_oracle_from_product, to_float32_state, to_float32_randomness)

arch = sys.argv[1] if len(sys.argv) > 1 else "celeba64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
std = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 5
gan, st, reals, rng = T._make(arch, B, std, seed=seed)
hp = dict(S.DEFAULT_HP, global_batch_size=B)
for mod in (gan.generator, gan.discriminator):
    mod.store.ensure_opt_state()
for it in range(steps):
    if it:
        sync_oracle_from_product(st, gan)
    rnd = S.draw_randomness(arch, B, rng, np.float64)
    r = rng.uniform(-1, 1, size=reals.shape)
    m_old = {k: [a.astype(np.float64) for a in product_slots(mod, "m")] for k, mod in (("g", gan.generator), ("d", gan.discriminator))}
    st32, _, aux32 = S.train_on_batch(to_float32_state(st), r.astype(np.float32), to_float32_randomness(rnd), hp)
    st, met, aux = S.train_on_batch(st, r, rnd, hp)
    gan.train_on_batch(r.astype(np.float32), randomness=rnd)
    for key, mod in (("d", gan.discriminator), ("g", gan.generator)):
        prod = [(a.astype(np.float64) - 0.9 * b) / (1.0 - float(np.float32(0.9))) for a, b in zip(product_slots(mod, "m"), m_old[key])]
        ref, ref32 = oracle_grad_list(aux[f"{key}_grads"]), oracle_grad_list(aux32[f"{key}_grads"])
        for i, (a, b, c) in enumerate(zip(prod, ref, ref32)):
            b, c = np.asarray(b, np.float64).reshape(a.shape), np.asarray(c, np.float64).reshape(a.shape)
            print(f"step {it} {key}{i:02d} {str(a.shape):22s} |ref| {np.linalg.norm(b):.2e}  HIP rel-L2 {rel_l2(a, b):.1e} 1-cos {1 - cosine(a, b):.0e}   "
                  f"f32 oracle rel-L2 {rel_l2(c, b):.1e} 1-cos {1 - cosine(c, b):.0e}", flush=True)
