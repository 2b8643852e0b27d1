"""Statistical loss / SWD curve bands of the CPU oracle on a real training run (tests/golden/curve_mnist.npz).

`north_star`: "loss/FID curves within tolerance of the CPU reference".  The reference (TensorFlow) cannot run here and RNG streams
cannot be shared with it anyway, so curve parity is STATISTICAL: the oracle's torch-CPU trainer (oracle/torch_ref.py,
wgan.py:86-114 step by step) trains the MNIST-architecture BlurredWGANGP for 400 steps at batch 64 on a seeded synthetic image
distribution (tests/golden/synth_data.py), with its own RNG, from K different seeds (weights, latents, alpha, dropout masks, batch
order).  Per window of 25 steps the file stores, per seed, the mean of disc_loss / gen_loss / gp_term / real_scores / fake_scores
(Q6 average) and, every 100 steps, SWD(fakes, reals) by the pinned sliced-Wasserstein code (callbacks.py:138-206 feeders'
preprocessing).  tests/test_curve_gpu.py trains the HIP product the same way with ITS OWN RNG and asserts every window inside
mean +- max(FACTOR x seed-to-seed spread, FLOOR).  A held-out oracle seed (must pass) and two deliberately wrong oracles --
gp_coefficient 5 instead of 10, and no [B]-vector loss quirk (Q1) -- (must fail) are evaluated here and stored, so the band is
shown to accept an independent correct run and to reject a wrong algorithm.

Run from the repo root (about 6 minutes on 4 threads):   python tests/golden/make_curve_golden.py
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "blurred-gan_amd"))

import synth_data as SD                                   # noqa: E402
from oracle import torch_ref as TR, models as M           # noqa: E402
import sliced_wasserstein as sw                           # noqa: E402  (the build's pinned rewrite: tests/test_metrics_cpu.py)

ARCH, BATCH, STEPS, WINDOW, SWD_EVERY, SWD_N = "mnist", 64, 400, 25, 100, 512
SEEDS = [101, 202, 303, 404, 505]
HELD_OUT = 909
METRICS = ["disc_loss", "gen_loss", "gp_term", "real_scores", "fake_scores"]
FACTOR, FLOORS = 4.0, dict(disc_loss=0.25, gen_loss=0.5, gp_term=0.1, real_scores=0.5, fake_scores=0.5, swd=25.0)
DATA_SEED, DATA_N = 2024, 4096


def swd_value(fakes, reals, seed):
    api = sw.API((len(reals), reals.shape[1], reals.shape[2], 3), seed=seed)
    api.begin("reals"); api.feed("reals", SD.to_swd_input(reals)); api.end("reals")
    api.begin("fakes"); api.feed("fakes", SD.to_swd_input(fakes)); res = api.end("fakes")
    return float(res[-1])


def run(seed, data, hp_extra=None, vector_quirk=True, log=None, lrelu_alpha=None, sigma=None, tag="ok"):
    cache = os.path.join(os.environ.get("CURVE_CACHE", "/tmp/curve"), f"run_{tag}_{seed}.npz")
    if os.path.exists(cache):
        z = np.load(cache)
        return {m: z[f"win_{m}"] for m in METRICS}, z["swd"], {m: z[f"step_{m}"] for m in METRICS}
    out = _run(seed, data, hp_extra, vector_quirk, log, lrelu_alpha, sigma)
    os.makedirs(os.path.dirname(cache), exist_ok=True)
    np.savez(cache, swd=out[1], **{f"win_{m}": out[0][m] for m in METRICS}, **{f"step_{m}": out[2][m] for m in METRICS})
    return out


def _run(seed, data, hp_extra, vector_quirk, log, lrelu_alpha, sigma):
    from oracle import np_ops as O
    torch.manual_seed(seed)
    saved_alpha = O.LRELU_ALPHA
    if lrelu_alpha is not None:
        O.LRELU_ALPHA = lrelu_alpha
    try:
        return _run_inner(seed, data, hp_extra, vector_quirk, log, sigma)
    finally:
        O.LRELU_ALPHA = saved_alpha


def _run_inner(seed, data, hp_extra, vector_quirk, log, sigma):
    hp = dict(global_batch_size=BATCH)
    hp.update(hp_extra or {})
    tr = TR.TorchTrainer(ARCH, seed=seed, std=SD.sigma_schedule(0), hp=hp)
    gen = torch.Generator().manual_seed(seed + 1)
    per_step = {m: [] for m in METRICS}
    swd = []
    if not vector_quirk:                                   # Q1 off: the loss the paper means (a scalar), for the rejection check
        orig = TR.discriminator_step_grads

        def scalar_loss(st_t, reals, rnd, hp_):
            out, met, fakes = orig(st_t, reals, rnd, hp_)
            B = reals.shape[0]
            # the [B]-vector sum multiplies the Wasserstein + penalty gradients by B; remove that factor (drift term is 1e-4: ignored)
            return [{k: v / B for k, v in d.items()} for d in out], met, fakes
        TR.discriminator_step_grads = scalar_loss
    try:
        for step, reals in enumerate(SD.batches(data, BATCH, STEPS, seed + 2)):
            tr.std = SD.sigma_schedule(step) if sigma is None else sigma
            met = tr.train_on_batch(torch.from_numpy(reals), tr.draw(BATCH, gen))
            met["fake_scores"] = 0.5 * (met["fake_scores"] + met["fake_scores_g"])        # Q6
            for m in METRICS:
                per_step[m].append(met[m])
            if (step + 1) % SWD_EVERY == 0:
                with torch.no_grad():
                    z = torch.rand(SWD_N, M.LATENT[ARCH], generator=gen)
                    fakes = TR.forward(tr.gspec, tr.g, z, training=False).numpy()
                swd.append(swd_value(fakes, data[:SWD_N], seed=7))
            if log and (step + 1) % 100 == 0:
                log(f"  seed {seed} step {step + 1}: disc {met['disc_loss']:.3f} gen {met['gen_loss']:.3f} gp {met['gp_term']:.3f} swd {swd[-1]:.1f}")
    finally:
        if not vector_quirk:
            TR.discriminator_step_grads = orig
    win = {m: np.asarray(per_step[m], np.float64).reshape(-1, WINDOW).mean(1) for m in METRICS}
    return win, np.asarray(swd, np.float64), {m: np.asarray(v, np.float64) for m, v in per_step.items()}


def bands(per_seed_win, per_seed_swd):
    out = {}
    for m in METRICS:
        a = np.stack([w[m] for w in per_seed_win])
        out[m] = (a.mean(0), a.std(0, ddof=1))
    s = np.stack(per_seed_swd)
    out["swd"] = (s.mean(0), s.std(0, ddof=1))
    return out


def violations(band, win, swd, factor=FACTOR, floors=FLOORS):
    """[(metric, window, value, mean, half-width)] outside the band -- the criterion tests/test_curve_gpu.py applies."""
    bad = []
    for m in METRICS + ["swd"]:
        mean, sd = band[m]
        vals = swd if m == "swd" else win[m]
        half = np.maximum(factor * sd, floors[m])
        for i in np.nonzero(np.abs(vals - mean) > half)[0]:
            bad.append((m, int(i), float(vals[i]), float(mean[i]), float(half[i])))
    return bad


def main():
    torch.set_num_threads(int(os.environ.get("CURVE_THREADS", "4")))
    data = SD.blob_dataset(DATA_N, 28, 1, DATA_SEED)
    t0 = time.time()
    wins, swds, steps = [], [], []
    for s in SEEDS:
        w, sv, ps = run(s, data, log=print)
        wins.append(w); swds.append(sv); steps.append(ps)
        print(f"seed {s} done after {time.time() - t0:.0f} s")
    band = bands(wins, swds)
    w, sv, _ = run(HELD_OUT, data, log=print)
    held = violations(band, w, sv)
    w5, sv5, _ = run(HELD_OUT, data, hp_extra=dict(gp_coefficient=5.0), log=print)
    wrong_gp = violations(band, w5, sv5)
    wq, svq, _ = run(HELD_OUT, data, vector_quirk=False, log=print)
    wrong_q1 = violations(band, wq, svq)
    print("held-out oracle seed: violations", held)
    print("gp_coefficient=5 oracle: violations", len(wrong_gp), wrong_gp[:4])
    print("scalar-loss (no Q1) oracle: violations", len(wrong_q1), wrong_q1[:4])
    assert not held, "the band rejects an independent run of the same oracle: widen FACTOR / FLOORS"
    assert wrong_gp and wrong_q1, "the band accepts a wrong algorithm: it has no power"
    out = dict(arch=ARCH, batch=BATCH, steps=STEPS, window=WINDOW, swd_every=SWD_EVERY, swd_n=SWD_N, seeds=np.asarray(SEEDS),
               data_seed=DATA_SEED, data_n=DATA_N, factor=FACTOR, metrics=np.asarray(METRICS),
               floors=np.asarray([FLOORS[m] for m in METRICS + ["swd"]]),
               held_out_violations=len(held), wrong_gp_violations=len(wrong_gp), wrong_q1_violations=len(wrong_q1))
    for m in METRICS + ["swd"]:
        out[f"{m}_mean"], out[f"{m}_sd"] = band[m]
    for m in METRICS:
        out[f"{m}_per_seed"] = np.stack([w_[m] for w_ in wins])
    out["swd_per_seed"] = np.stack(swds)
    path = os.path.join(HERE, "curve_mnist.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, f"({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
