"""Mirror of reference wgan.py: ``WGAN``, ``WGANGP``, ``gradient_penalty``, ``TrainingConfig`` with the same
constructor, attributes and method names, executing on the HIP kernels (no tape, no torch autograd).

Step order, training flags and loss algebra follow wgan.py:86-172, 234-285 including its quirks
(SURVEY.md 8a Q1, Q2, Q4, Q6); each quirk that changes numbers is a named constructor switch.
"""
from __future__ import annotations

import math
import numbers
import os
from contextlib import contextmanager
from dataclasses import dataclass
from typing import List

import numpy as np
import torch

from . import dist, ops, program
from .gaussian_blur import Variable
from .layers import Sequential, get_seed
from .utils import JsonSerializable, ParseableFromCommandLine

ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-7       # tf.keras.optimizers.Adam defaults (wgan.py:56)


_KEY_ENV = ("BGAN_NO_FUSED_BLUR3", "BGAN_NO_FOLD_MANY", "BGAN_NO_FUSED_BN_STATS", "BG_BLUR_NO_ROWS", "BG_BLUR_NO_PANEL", "BG_BLUR_PANEL16", "BG_WGRAD_NO_STRIP")


def _env_switches():
    """Dispatch switches that the engine / the library read per call and that change a step's launch list: part of the step-program
    key, so flipping one mid-run records a new program instead of replaying a stale launch list."""
    return tuple(os.environ.get(k) for k in _KEY_ENV)


def _plain(v):
    """Hyper-parameter values as plain Python scalars (np.float32(1e-3) and 1e-3 must give the same program key)."""
    if isinstance(v, (bool, np.bool_)):
        return bool(v)
    if isinstance(v, (numbers.Integral, np.integer)):
        return int(v)
    if isinstance(v, (numbers.Real, np.floating)):
        return float(v)
    return v


@dataclass
class TrainingConfig(JsonSerializable, ParseableFromCommandLine):
    """Parameters related to the training configuration of the Model (wgan.py:19-25)."""
    log_dir: str = "results/log"
    checkpoint_dir: str = "results/log/checkpoints"
    save_image_summaries_interval: int = 50


class Mean:
    """tf.keras.metrics.Mean stand-in (host side)."""

    def __init__(self, name, dtype=None):
        self.name = name
        self.reset_states()

    def reset_states(self):
        self.total, self.count = 0.0, 0

    def __call__(self, value, weight=1):
        self.total += float(value) * weight
        self.count += weight

    update_state = __call__

    def result(self):
        return self.total / self.count if self.count else 0.0


class _Adam:
    """Keras Adam bookkeeping; the update itself is one fused launch over the model's flat buffer."""

    def __init__(self, learning_rate=0.001):
        self.learning_rate = learning_rate
        self.iterations = 0

    def apply(self, store):
        self.iterations += 1
        t = self.iterations
        lr_t = float(self.learning_rate) * math.sqrt(1.0 - ADAM_B2 ** t) / (1.0 - ADAM_B1 ** t)
        if program.active() is not None:        # step program: the recorded launch takes lr_t of the NEXT iteration from a slot
            program.active().bind_adam(self, ADAM_B1, ADAM_B2)
        ops.adam(store.theta[:store.n_train], store.m[:store.n_train], store.v[:store.n_train],
                 store.grad[:store.n_train], lr_t, ADAM_B1, ADAM_B2, ADAM_EPS)
        store.tr_dirty = True


class _SummaryWriter:
    """Very small stand-in for tf.summary's file writer: scalars appended to <log_dir>/scalars.jsonl."""

    def __init__(self, log_dir):
        self.log_dir = log_dir
        self._fh = None

    @contextmanager
    def as_default(self):
        yield self

    def scalar(self, name, value, step=None):
        import json
        if self._fh is None:
            os.makedirs(self.log_dir, exist_ok=True)
            self._fh = open(os.path.join(self.log_dir, "scalars.jsonl"), "a")
        self._fh.write(json.dumps({"name": name, "value": float(value), "step": None if step is None else int(step)}) + "\n")
        self._fh.flush()


class WGAN:
    """Wasserstein GAN (wgan.py:28-231)."""

    @dataclass
    class HyperParameters(JsonSerializable, ParseableFromCommandLine):
        """Dataclass containing the hyperparameters of the Model (wgan.py:34-43)."""
        learning_rate: float = 0.001
        d_steps_per_g_step: int = 1
        batch_size: int = 32
        global_batch_size: int = 32
        optimizer: str = "adam"

    uses_gradient_penalty = False

    def __init__(self, generator: Sequential, discriminator: Sequential, hyperparams: "WGAN.HyperParameters",
                 config: TrainingConfig, *args, reproduce_vector_loss_quirk: bool = True, sync_metrics: bool = True,
                 sync_batchnorm: bool = True, merge_critic_passes: bool = True, gp_zero_norm_guard: bool = False,
                 merge_gp_filter_gradients: bool = True, step_replay: bool = True, persistent_input: bool = False,
                 **kwargs):
        self.hparams = hyperparams
        if dist.world_size() > 1 and int(hyperparams.global_batch_size) != int(hyperparams.batch_size) * dist.world_size():
            import warnings
            # Q2: the reference never writes the true global batch into the hyper-parameters; under data parallelism the loss
            # scale 1/global_batch_size and the penalty's global mean must agree, so say so instead of training a different model
            warnings.warn(f"global_batch_size={hyperparams.global_batch_size} but {dist.world_size()} replicas x batch_size="
                          f"{hyperparams.batch_size}: the Wasserstein / penalty gradient ratio differs from the single-device "
                          "step at the global batch (set global_batch_size = batch_size * replicas)")
        if str(self.hparams.optimizer).lower() != "adam":
            raise NotImplementedError("only the reference's default optimizer 'adam' is implemented (wgan.py:43,56)")
        self.generator = generator
        self.generator.build()
        self.generator.optimizer = _Adam(self.hparams.learning_rate)
        self.discriminator = discriminator
        self.discriminator.build()
        self.discriminator.optimizer = _Adam(self.hparams.learning_rate)
        self.d_steps_per_g_step = self.hparams.d_steps_per_g_step
        self.batch_size = None
        self.config = config
        self.summary_writer = _SummaryWriter(config.log_dir)
        self.n_img = Variable(0, name="n_img", dtype=int)
        self.n_batches = Variable(0, name="n_batches", dtype=int)
        self.real_scores_metric = Mean("real_scores")
        self.fake_scores_metric = Mean("fake_scores")
        self.gen_loss_metric = Mean("gen_loss")
        self.disc_loss_metric = Mean("disc_loss")
        self.optimizer = "unused"
        self.stop_training = False
        self.images = None
        # --- build-side switches (documented in DESIGN.md)
        self.reproduce_vector_loss_quirk = reproduce_vector_loss_quirk
        # Metrics live in one 16-float device buffer (critic step [0:8], generator step [8:12]).  train_on_batch reads it ONCE,
        # after the last kernel of the step has been enqueued (asynchronous copy into pinned memory + event wait): the one
        # host<->device synchronisation of a step, where Keras' train_on_batch converts its metric tensors (wgan.py:114).
        # sync_metrics=False skips even that (the returned metrics are then zeros): a measuring aid, never used by bench.py.
        self.sync_metrics = sync_metrics
        self._defer_metrics = False
        self._met_host = None
        self._met_event = None
        self.sync_batchnorm = sync_batchnorm  # DP: generator BN statistics over the global batch (False = per replica)
        # critic step: [fakes; reals] and x-hat in one 3B-sample forward / backward (False: two passes, as the reference orders them)
        self.merge_critic_passes = merge_critic_passes and not os.environ.get("BGAN_NO_MERGED_CRITIC")
        # the penalty's second-order filter gradients ride in the merged pass's own filter-gradient launches
        # (engine.Net.gp_second_order_merged); needs the merged critic pass
        self.merge_gp_filter_gradients = merge_gp_filter_gradients and not os.environ.get("BGAN_NO_MERGED_GP_WGRAD")
        # a sample whose critic input-gradient is exactly zero makes the penalty's second-order seed (n-1)/n * g = NaN, in the
        # reference too (tf.norm's gradient at 0); True takes the subgradient 0 for that sample instead
        self.gp_zero_norm_guard = gp_zero_norm_guard
        # train_on_batch records the launch list of its discriminator_step / generator_step once and replays it with one call
        # into the library (program.py, include/bgan.h bg_dstep / bg_gstep); False or BGAN_NO_STEP_REPLAY=1: every step eager
        self.step_replay = step_replay and program.enabled_by_env()
        self._programs = program.StepPrograms()
        self._reals_stage = None
        self.persistent_input = bool(persistent_input)
        self._rng_seed = get_seed()
        self._rng_off = 0
        self._bufs = {}
        self._injected = None

    # ------------------------------------------------------------------ small helpers
    @property
    def device(self):
        return self.generator.store.device

    @property
    def latent_size(self):
        return self.generator.input_shape[-1]

    @property
    def metrics(self) -> List[Mean]:
        return [self.real_scores_metric, self.fake_scores_metric, self.gen_loss_metric, self.disc_loss_metric]

    @property
    def metrics_names(self):
        return ["loss"] + [m.name for m in self.metrics]

    def reset_metrics(self):
        for m in self.metrics:
            m.reset_states()

    def _buf(self, name, shape, dtype=torch.float32):
        key = (name, tuple(shape), dtype)
        b = self._bufs.get(key)
        if b is None:
            b = self._bufs[key] = torch.empty(*shape, dtype=dtype, device=self.device)
        return b

    def _uniform(self, name, shape):
        """tf.random.uniform stand-in (wgan.py:118,237): own counter-based stream, seed + rank."""
        out = self._buf(name, shape)
        ops.uniform(out, self._rng_seed + 7919 * dist.rank(), counter=(self, "_rng_off"))
        return out

    def _inj(self, key):
        return None if self._injected is None else self._injected.get(key)

    def _vec_scale(self, B):
        """Q1: the reference adds a [B] drift vector to the scalar loss, and GradientTape sums it, which
        multiplies the Wasserstein and GP gradients by the (global) batch size (wgan.py:279-285)."""
        return float(B * dist.world_size()) if (self.uses_gradient_penalty and self.reproduce_vector_loss_quirk) else 1.0

    # ------------------------------------------------------------------ reference API
    def train_on_batch(self, reals, *args, randomness=None, **kwargs):
        """wgan.py:86-114.  ``randomness`` (build-side, for parity tests) injects the step's random inputs:
        dict(z_d, z_g, alpha, mask_fake, mask_real)."""
        self.reset_metrics()
        reals = self._as_device(reals)
        self.batch_size = int(reals.shape[0])
        self._injected = randomness
        self._defer_metrics = True
        replay = self.step_replay and randomness is None and self._replayable()
        try:
            if replay:
                # a recorded program holds device addresses: the batch goes through ONE persistent staging buffer
                reals = self._stage_reals(reals)
                stream = ops._stream()
                with ops.trace_range("d_step"):
                    disc_loss, self.images = self._programs.run(self._step_key("d", reals), lambda: self.discriminator_step(reals),
                                                                stream, self._exit_state)
                    if self._programs.last_was_replay:
                        self._after_d_step()
                g_ran = int(self.n_batches) % self.d_steps_per_g_step == 0
                if g_ran:
                    with ops.trace_range("g_step"):
                        self._programs.run(self._step_key("g", reals), self.generator_step, stream, self._exit_state)
            else:
                with ops.trace_range("d_step"):
                    disc_loss, self.images = self.discriminator_step(reals)
                g_ran = int(self.n_batches) % self.d_steps_per_g_step == 0
                if g_ran:
                    with ops.trace_range("g_step"):
                        self.generator_step()
        finally:
            self._injected = None
            self._defer_metrics = False
        if self.sync_metrics:
            m = self._read_metrics()
            self._record_d_metrics(m[:8])
            if g_ran:
                self._record_g_metrics(m[8:12])
        self.log_image_summaries()
        self.n_img.assign_add(self.batch_size)
        self.n_batches.assign_add(1)
        return self._organize_metrics()

    def _after_d_step(self):
        """Host-side bookkeeping of discriminator_step that a replayed step program has to repeat (subclasses)."""

    # ---- step programs (program.py)
    def _replayable(self):
        G, D = self.generator.net(), self.discriminator.net()
        return self.device.type == "cuda" and G.capture_branches is None and D.capture_branches is None

    def _stage_reals(self, reals):
        """The batch a recorded program reads.  A program holds device addresses, so by default the batch goes through ONE
        persistent staging buffer (a copy of the batch per step, 4-10 us).  ``persistent_input = True`` (constructor keyword or
        attribute) is the caller's promise that its batches arrive in device buffers that stay allocated and are refilled in
        place (a prefetcher with one or two device buffers): programs are then recorded on those buffers directly -- the address
        is part of the program key -- and no copy is made."""
        if self.persistent_input:
            return reals
        # one staging buffer PER BATCH SHAPE, kept for the life of the model: recorded programs hold its address, so a buffer that
        # was re-allocated on every shape switch (the partial last batch of an epoch) would strand the programs of both shapes at
        # each switch -- two eager steps and a recording per step kind, every epoch
        if self._reals_stage is None:
            self._reals_stage = {}
        key = (tuple(reals.shape), reals.dtype, reals.device)
        st = self._reals_stage.get(key)
        if st is None:
            st = self._reals_stage[key] = torch.empty_like(reals)
        if st.data_ptr() != reals.data_ptr():
            st.copy_(reals, non_blocking=True)
        return st

    def _step_key(self, kind, reals):
        """Everything that shapes the launch list of a step or is baked into its kernel arguments."""
        G, D = self.generator.net(), self.discriminator.net()
        hp = tuple(sorted((k, _plain(v)) for k, v in vars(self.hparams).items() if isinstance(v, (numbers.Number, np.generic, str, bool))))
        return (kind, tuple(reals.shape), reals.data_ptr(), D.blur_n_taps(), G.store.tr_dirty, D.store.tr_dirty, self.merge_critic_passes,
                self.merge_gp_filter_gradients, self.sync_batchnorm, self.gp_zero_norm_guard, self.reproduce_vector_loss_quirk,
                self.sync_metrics, dist.collectives_active(), dist.world_size(), G.fuse_bn_stats, D.fuse_bn_stats,
                G.store.n_train, D.store.n_train, G.bn_bwd_read_y, _env_switches(), hp)

    def _exit_state(self):
        G, D = self.generator.net(), self.discriminator.net()
        return [(G.store, "tr_dirty", G.store.tr_dirty), (D.store, "tr_dirty", D.store.tr_dirty)]

    def _metrics_dev(self):
        return self._buf("step_metrics", (16,))

    def _read_metrics(self):
        """ONE device->host read of the step's metric buffer: asynchronous copy into pinned memory behind everything enqueued
        so far, then wait for that copy only."""
        dev = self._metrics_dev()
        if self._met_host is None:
            self._met_host = torch.empty(16, dtype=torch.float32, pin_memory=dev.is_cuda)
            self._met_event = torch.cuda.Event() if dev.is_cuda else None
        self._met_host.copy_(dev, non_blocking=True)
        if self._met_event is not None:
            self._met_event.record()
            self._met_event.synchronize()
        return self._met_host.tolist()

    def _record_d_metrics(self, m):
        self.fake_scores_metric(m[0])
        self.real_scores_metric(m[1])
        self.disc_loss_metric(m[2])
        self._record_gp_metrics(m)

    def _record_g_metrics(self, m):
        self.fake_scores_metric(m[0])                      # Q6: second update of the same Mean
        self.gen_loss_metric(m[1])

    def _as_device(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        return x.to(self.device, torch.float32).contiguous()

    def latents_batch(self):
        assert self.batch_size is not None
        return self._uniform("latents", (self.batch_size, self.latent_size))

    def generate_samples(self, latents=None, training=False):
        if latents is None:
            if self.batch_size is None:
                self.batch_size = self.hparams.batch_size
            latents = self.latents_batch()
        latents = self._as_device(latents)
        G = self.generator.net()
        ctx = G.context(int(latents.shape[0]), "g")
        return G.forward(ctx, latents, training=training)

    # ---- discriminator
    def discriminator_loss(self, reals, fakes, real_scores, fake_scores):
        """wgan.py:128-130 (value only; the training step uses the fused device path)."""
        return (fake_scores - real_scores).sum() * (1.0 / self.hparams.global_batch_size)

    def discriminator_step(self, reals):
        """wgan.py:132-151."""
        reals = self._as_device(reals)
        B = int(reals.shape[0])
        self.batch_size = B
        hp = self.hparams
        G, D = self.generator.net(), self.discriminator.net()
        z = self._inj("z_d")
        z = self._as_device(z) if z is not None else self._uniform("z_d", (B, self.latent_size))
        fakes = G.forward(G.context(B, "g_dstep"), z, training=False)           # Q4: inference BN in the D-step
        masks = None
        if self._inj("mask_fake") is not None:
            masks = [torch.cat([self._as_mask(a), self._as_mask(b)], 0)
                     for a, b in zip(self._inj("mask_fake"), self._inj("mask_real"))]
        inv_gbs = 1.0 / float(hp.global_batch_size)
        gp_c = float(getattr(hp, "gp_coefficient", 0.0)) if self.uses_gradient_penalty else 0.0
        e_d = float(getattr(hp, "e_drift", 0.0)) if self.uses_gradient_penalty else 0.0
        met = self._metrics_dev()[:8]
        vs = self._vec_scale(B)
        store = self.discriminator.store
        store.ensure_opt_state()
        red = dist.GradReducer(store.grad, store.n_train)       # buckets go out while the backward is still running
        seed = self._rng_seed + 104729 * (dist.rank() + 1)
        merged = (self.uses_gradient_penalty and self.merge_critic_passes and all(st.bn is None for st in D.stages))
        if merged:
            # [fakes; reals] (training=True: Dropout) and x-hat (training=False) share every weight: ONE 3B-sample pass forward
            # and ONE backward, the Dropout masks covering the first 2B samples only (bg_epilogue.keep_elems).  The seeds of
            # the backward do not depend on the penalty (d loss / d scores is +-1/gbs and the drift term), so the x-hat rows
            # ride along with seed 1 and come out as the zeta_i the second-order pass needs; weight gradients use rows [0, 2B),
            # the image gradient is taken for rows [2B, 3B) only.
            a = self._inj("alpha")
            a = self._as_device(a).view(B) if a is not None else self._uniform("alpha", (B,))
            c3 = D.context(3 * B, "fr3", drop_rows=2 * B)
            # x-hat = reals + a (fakes - reals) (wgan.py:239) joins the batch inside forward: behind the blur it is formed on the fly
            # by the one launch that blurs all three slices (bg_blur3_lerp_nhwc_f32), else bg_lerp_f32 writes it to the buffer
            s3 = D.forward(c3, [fakes, reals], training=True, masks=masks, seed=seed, lerp_alpha=a,
                           lerp_out=self._buf("xhat", tuple(reals.shape))).view(3 * B)
            fs, rs = s3[:B], s3[B:2 * B]
            ds3 = self._buf("ds3", (3 * B,))
            ops.fill(ds3[2 * B:], 1.0)
            ops.wgangp_d_loss(fs, rs, None, inv_gbs, gp_c, e_d, vs, ds3[:B], ds3[B:2 * B], met)      # seeds only; metrics below
            one_wgrad = self.merge_gp_filter_gradients
            g = D.backward(c3, ds3.view(3 * B, 1), need_dx=True, need_dw=True, beta=0.0, scale=1.0, dw_rows=2 * B,
                           dx_rows=(2 * B, 3 * B), defer_conv_dw=one_wgrad)
            norms = ops.row_norm(g, self._buf("gp_norms", (B,)))
            ops.wgangp_d_loss(fs, rs, norms, inv_gbs, gp_c, e_d, vs, ds3[:B], ds3[B:2 * B], met)    # same seeds, full metrics
            coef = vs * float(hp.gp_coefficient) * 2.0 / float(B * dist.world_size())
            if one_wgrad:
                # delta-bar_0 goes into the x-hat rows of the pass's (blurred) input buffer, where the layer-1 filter gradient
                # reads it beside the activations of [fakes; reals]; g itself lives in those rows and is dead after the seed
                hat0 = c3.a0[2 * B:3 * B]
                if D.blur is not None:
                    gbar = ops.gp_seed(g, norms, coef, self._buf("gbar", tuple(g.shape)), self.gp_zero_norm_guard)
                    D.apply_blur(gbar, hat0)
                else:
                    ops.gp_seed(g, norms, coef, hat0.view(g.shape), self.gp_zero_norm_guard)      # g is the net's own input-gradient buffer
                D.gp_second_order_merged(c3, 2 * B, 3 * B, reducer=red)
            else:
                gbar = ops.gp_seed(g, norms, coef, self._buf("gbar", tuple(g.shape)), self.gp_zero_norm_guard)
                v0 = D.apply_blur(gbar, self._buf("v0", tuple(g.shape))) if D.blur is not None else gbar
                D.gp_second_order(c3.rows(2 * B, 3 * B), v0, reducer=red)
        else:
            cfr = D.context(2 * B, "fr")
            s2 = D.forward(cfr, [fakes, reals], training=True, masks=masks, seed=seed).view(2 * B)
            fs, rs = s2[:B], s2[B:]
            norms = g = None
            if self.uses_gradient_penalty:
                norms, g = self._gp_first_order(reals, fakes)
            ds2 = self._buf("ds2", (2 * B,))
            ops.wgangp_d_loss(fs, rs, norms, inv_gbs, gp_c, e_d, vs, ds2[:B], ds2[B:], met)
            D.backward(cfr, ds2.view(2 * B, 1), need_dx=False, need_dw=True, beta=0.0, scale=1.0,
                       reducer=None if self.uses_gradient_penalty else red)
            if self.uses_gradient_penalty:
                # d/dW of vec_scale * gp_coefficient * mean_global((n-1)^2): seed carries the whole factor
                coef = vs * float(hp.gp_coefficient) * 2.0 / float(B * dist.world_size())
                gbar = ops.gp_seed(g, norms, coef, self._buf("gbar", tuple(g.shape)), self.gp_zero_norm_guard)
                chat = D.context(B, "hat")
                v0 = D.apply_blur(gbar, self._buf("v0", tuple(g.shape))) if D.blur is not None else gbar
                D.gp_second_order(chat, v0, reducer=red)
        red.finish()
        self.discriminator.optimizer.apply(store)
        disc_loss = None              # inside train_on_batch the value is read with the rest of the step's metrics
        if self.sync_metrics and not self._defer_metrics:
            m = self._read_metrics()
            self._record_d_metrics(m[:8])
            disc_loss = m[2]
        return disc_loss, (fakes, reals)

    def _record_gp_metrics(self, m):
        pass

    @staticmethod
    def _as_mask(a):
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(a)
        return a.to(torch.uint8)

    def _gp_first_order(self, reals, fakes):
        """wgan.py:234-245 on the device: x_hat, critic forward (training=False), gradient w.r.t. x_hat
        (through blur^T), per-sample L2 norms.  Leaves zeta_i in the 'hat' context for the second order."""
        B = int(reals.shape[0])
        D = self.discriminator.net()
        a = self._inj("alpha")
        a = self._as_device(a).view(B) if a is not None else self._uniform("alpha", (B,))
        xhat = ops.lerp(reals, fakes, a, self._buf("xhat", tuple(reals.shape)))
        chat = D.context(B, "hat")
        D.forward(chat, xhat, training=False)
        ones = ops.fill(self._buf("ones", (B, 1)), 1.0)
        g = D.backward(chat, ones, need_dx=True, need_dw=False)
        norms = ops.row_norm(g, self._buf("gp_norms", (B,)))
        return norms, g

    # ---- generator
    def generator_loss(self, fake_scores):
        """wgan.py:155-157."""
        return -fake_scores.sum() * (1.0 / self.hparams.global_batch_size)

    def generator_step(self):
        """wgan.py:159-172."""
        B = self.batch_size
        G, D = self.generator.net(), self.discriminator.net()
        z = self._inj("z_g")
        z = self._as_device(z) if z is not None else self._uniform("z_g", (B, self.latent_size))
        cg = G.context(B, "g")
        G.sync_bn = self.sync_batchnorm
        fakes = G.forward(cg, z, training=True)
        chat = D.context(B, "hat")
        s = D.forward(chat, fakes, training=False).view(B)
        ds = self._buf("ds_g", (B,))
        met = self._metrics_dev()[8:12]
        ops.wgan_g_loss(s, 1.0 / float(self.hparams.global_batch_size), ds, met)
        dfakes = D.backward(chat, ds.view(B, 1), need_dx=True, need_dw=False)
        store = self.generator.store
        store.ensure_opt_state()
        red = dist.GradReducer(store.grad, store.n_train)
        G.backward(cg, dfakes, need_dx=False, need_dw=True, beta=0.0, scale=1.0, reducer=red)
        red.finish()
        self.generator.optimizer.apply(store)
        if self.sync_metrics and not self._defer_metrics:
            m = self._read_metrics()
            self._record_g_metrics(m[8:12])
            return m[1 + 8]
        return None

    # ---- bookkeeping / Keras surface
    def log_image_summaries(self):
        """wgan.py:176-180: image summaries are a TensorBoard feature; not written by this build."""
        return None

    def _organize_metrics(self) -> List[float]:
        """wgan.py:182-200: [0.0] + metric results in metrics_names order."""
        by_name = {m.name: m for m in self.metrics}
        assert len(by_name) == len(self.metrics), "duplicate metric names"
        return [0.0] + [by_name[n].result() for n in self.metrics_names if n != "loss"]

    @contextmanager
    def record_image_summaries(self):
        yield

    def summary(self):
        print("Discriminator:")
        self.discriminator.summary()
        print("Generator:")
        self.generator.summary()
        print(f"Total params: {self.count_params():,}")

    def count_params(self):
        return self.discriminator.count_params() + self.generator.count_params()

    def save_weights(self, filepath, overwrite=True, save_format=None):
        """wgan.py:229-231."""
        self.discriminator.save_weights(filepath + "_discriminator", overwrite, save_format)
        self.generator.save_weights(filepath + "_generator", overwrite, save_format)

    def fit(self, x, y=None, epochs=1, initial_epoch=0, callbacks=None, steps_per_epoch=None, verbose=0):
        """The slice of ``tf.keras.Model.fit`` the reference demos rely on (demo_mnist.py:187-206): iterate
        batches, call the Keras callback hooks around ``train_on_batch``, honour ``stop_training``."""
        callbacks = list(callbacks or [])
        for cb in callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(self)
            else:
                cb.model = self
        call = lambda name, *a: [getattr(cb, name)(*a) for cb in callbacks if hasattr(cb, name)]
        call("on_train_begin", {})
        history = []
        for epoch in range(int(initial_epoch), int(epochs)):
            call("on_epoch_begin", epoch, {})
            logs = {}
            for batch, reals in enumerate(x):
                if steps_per_epoch is not None and batch >= steps_per_epoch:
                    break
                size = int(reals.shape[0])
                call("on_batch_begin", batch, {"batch": batch, "size": size})
                values = self.train_on_batch(reals)
                logs = {"batch": batch, "size": size}
                logs.update(dict(zip(self.metrics_names, values)))
                call("on_batch_end", batch, logs)
                if self.stop_training:
                    break
            history.append(dict(logs))
            call("on_epoch_end", epoch, logs)
            if self.stop_training:
                break
        call("on_train_end", {})
        return history


def gradient_penalty(discriminator, reals, fakes, alpha=None):
    """wgan.py:234-246 (value only): mean((||d D(x_hat)/d x_hat|| - 1)^2) on the HIP kernels."""
    D = discriminator.net()
    dev = D.device
    reals = reals.to(dev, torch.float32).contiguous()
    fakes = fakes.to(dev, torch.float32).contiguous()
    B = int(reals.shape[0])
    if alpha is None:
        alpha = ops.uniform(torch.empty(B, dtype=torch.float32, device=dev), get_seed() + 17, np.random.randint(1 << 30))
    xhat = ops.lerp(reals, fakes, alpha.to(dev, torch.float32).contiguous().view(B), torch.empty_like(reals))
    ctx = D.context(B, "gp_fn")
    D.forward(ctx, xhat, training=False)
    g = D.backward(ctx, ops.fill(torch.empty(B, 1, dtype=torch.float32, device=dev), 1.0), need_dx=True, need_dw=False)
    n = ops.row_norm(g, torch.empty(B, dtype=torch.float32, device=dev))
    # mean((n - 1)^2) by the loss kernel itself (metric slot 5 of bg_wgangp_d_loss), not by torch arithmetic
    zero = ops.fill(torch.empty(B, dtype=torch.float32, device=dev), 0.0)
    scratch, met = torch.empty(2, B, dtype=torch.float32, device=dev), torch.empty(8, dtype=torch.float32, device=dev)
    ops.wgangp_d_loss(zero, zero, n, 1.0, 0.0, 0.0, 1.0, scratch[0], scratch[1], met)
    return met[5]


class WGANGP(WGAN):
    """Wasserstein GAN with Gradient Penalty loss (wgan.py:249-285)."""

    @dataclass
    class HyperParameters(WGAN.HyperParameters):
        """Hyperparameters of a WGAN model with Gradient Penalty loss (wgan.py:255-261)."""
        e_drift: float = 1e-4
        gp_coefficient: float = 10.0

    uses_gradient_penalty = True

    def __init__(self, generator, discriminator, hyperparams, config, *args, **kwargs):
        super().__init__(generator, discriminator, hyperparams, config, *args, **kwargs)
        self.gp_term_metric = Mean("gp_term")
        self.norm_term_metric = Mean("norm_term")

    @property
    def metrics(self):
        return super().metrics + [self.gp_term_metric, self.norm_term_metric]

    def _record_gp_metrics(self, m):
        self.gp_term_metric(m[3])
        self.norm_term_metric(m[4])

    def discriminator_loss(self, reals, fakes, real_scores, fake_scores):
        """wgan.py:272-285 (value only): a [B] vector, as in the reference."""
        loss = super().discriminator_loss(reals, fakes, real_scores, fake_scores)
        gp_term = self.hparams.gp_coefficient * gradient_penalty(self.discriminator, reals, fakes)
        norm_term = self.hparams.e_drift * (fake_scores.abs().view(-1) + real_scores.abs().view(-1))
        return loss + gp_term + norm_term
