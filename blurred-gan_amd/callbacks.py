"""Mirror of reference callbacks.py: the Keras-callback controllers of the blur sigma and the periodic
host-side hooks.  Same class names, constructor arguments and hook methods; pure host code."""
from __future__ import annotations

import os
from typing import Dict

import numpy as np


class Callback:
    """tf.keras.callbacks.Callback stand-in: ``self.model`` is set by ``fit``."""

    def __init__(self):
        self.model = None

    def set_model(self, model):
        self.model = model


class ExponentialDecay:
    """tf.keras.optimizers.schedules.ExponentialDecay(staircase=False): init * rate ** (step / decay_steps),
    evaluated in float32 like TF."""

    def __init__(self, initial_learning_rate, decay_steps, decay_rate, staircase=False):
        self.initial, self.decay_steps, self.rate, self.staircase = float(initial_learning_rate), float(decay_steps), float(decay_rate), staircase

    def __call__(self, step):
        p = np.float32(float(step)) / np.float32(self.decay_steps)
        if self.staircase:
            p = np.floor(p)
        return float(np.float32(self.initial) * np.power(np.float32(self.rate), np.float32(p)))


class ExecuteEveryNExamplesCallback(Callback):
    """A hook that fires once per ``n`` training EXAMPLES rather than per batch (reference callbacks.py:12-43).

    Behaviour kept from the reference: examples are counted from the ``size`` entry of the batch logs; nothing fires before
    ``starting_from`` examples have been seen (a negative value makes the first batch due at once); at most ONE call is made per
    batch, so a batch that crosses several periods is caught up one batch at a time; the very first eligible batch fires
    (period index 0).  Subclasses implement ``function(batch, logs)``."""

    def __init__(self, n: int, starting_from: int = 0):
        super().__init__()
        self.period, self.starting_from = n, starting_from
        self.samples_seen = 0          # examples counted so far
        self.num_invocations = 0       # calls made so far == index of the next period that is owed a call

    def periods_elapsed(self) -> int:
        """Whole periods since ``starting_from`` (negative before it)."""
        return (self.samples_seen - self.starting_from) // self.period

    def on_batch_end(self, batch, logs: Dict):
        self.samples_seen += logs["size"]
        owed = self.samples_seen >= self.starting_from and self.periods_elapsed() >= self.num_invocations
        if owed:
            self.num_invocations += 1
            self.function(batch, logs)

    def function(self, batch, logs):
        raise NotImplementedError(f"{type(self).__name__} must implement function(batch, logs)")


class BlurDecayController(Callback):
    """callbacks.py:45-62.  Quirk Q5 reproduced: the schedule is evaluated at the BATCH index while
    ``decay_steps`` counts EXAMPLES; ``min_value`` is accepted and ignored, as in the reference."""

    def __init__(self, total_n_training_examples: int, max_value: float = 23.5, min_value=0.01):
        super().__init__()
        self.schedule = ExponentialDecay(float(max_value), decay_steps=total_n_training_examples / 10, decay_rate=0.96,
                                         staircase=False)

    def on_batch_begin(self, batch, logs):
        value = self.schedule(int(self.model.n_batches))
        self.model.std.assign(value)


class AdaptiveBlurController(Callback):
    """Score-balance controller of the blur (reference callbacks.py:65-135).

    It tracks an exponential moving average of fake / (real + fake) critic scores; once past ``warmup_n_batches`` it logs the raw
    and smoothed ratio and whether the game counts as balanced (smoothed ratio within ``threshold`` of 0.5), and while balanced it
    shrinks ITS OWN ``std`` by the smoothing factor, at most once per ``delay_between_modifications`` batches.  As in the reference
    the shrunken value is only logged ("would_modify") -- the assignment to the model's blur is commented out there
    (callbacks.py:102-103), so the model keeps the ``max_value`` set at train begin -- and training stops when the controller's
    value falls under ``min_value``."""

    delay_between_modifications = 100
    _TAG = "blur_controller/"

    def __init__(self, smoothing=0.99, warmup_n_batches=100, threshold=0.05, min_value=0.01, max_value=23.5):
        super().__init__()
        self.smoothing, self.warmup_n_batches, self.threshold, self.min_value = smoothing, warmup_n_batches, threshold, min_value
        self.std = float(max_value)
        self.score_ratio = 0.5                     # smoothed fake share of the scores; 0.5 = balanced
        self._last_modification_step = 0

    def on_train_begin(self, logs=None):
        self.model.std.assign(self.std)

    def gan_problem_is_stable(self) -> bool:
        return abs(self.score_ratio - 0.5) <= self.threshold

    def _scalars(self, **values):
        with self.model.summary_writer.as_default() as w:
            for name, v in values.items():
                w.scalar(self._TAG + name, v)

    def decrease_blur_std(self, batch: int) -> None:
        """One multiplicative step, unless the previous one was fewer than ``delay_between_modifications`` batches ago."""
        allowed = batch - self._last_modification_step >= self.delay_between_modifications
        if allowed:
            self.std *= self.smoothing
            self._last_modification_step = batch
        self._scalars(would_modify=int(allowed))

    def on_batch_end(self, batch, logs):
        fake, real = logs["fake_scores"], logs["real_scores"]
        ratio = fake / (real + fake)
        self.score_ratio = self.smoothing * self.score_ratio + (1 - self.smoothing) * ratio
        if batch < self.warmup_n_batches:
            return
        stable = self.gan_problem_is_stable()
        self._scalars(ratio=ratio, smoothed_ratio=self.score_ratio, stable=int(stable))
        if stable:
            self.decrease_blur_std(batch)
        if self.std < self.min_value:
            print("Reached the minimum STD. Training is complete.")
            self.model.stop_training = True


class GenerateSampleGridCallback(ExecuteEveryNExamplesCallback):
    """callbacks.py:209-236: an 8x8 grid of samples from fixed latents, written as PNG (PIL, no matplotlib)."""

    def __init__(self, log_dir: str, show_blurred_samples=True, every_n_examples=1000, also_save_files=True):
        self.log_dir = log_dir
        self.show_blurred_samples = show_blurred_samples
        super().__init__(n=every_n_examples)
        self.also_save_files = also_save_files
        self.latents = None

    def function(self, batch, logs):
        self.make_grid()

    def on_train_begin(self, logs: Dict):
        rng = np.random.default_rng(0)
        self.latents = rng.uniform(size=(64, self.model.generator.input_shape[-1])).astype(np.float32)

    def make_grid(self, *args):
        from .utils import normalize_images
        samples = self.model.generate_samples(self.latents, training=False)
        if self.show_blurred_samples and hasattr(self.model, "blur"):
            samples = self.model.blur(samples)
        s = normalize_images(samples).clamp(0, 1).cpu().numpy()
        n, h, w, c = s.shape
        grid = s[:64].reshape(8, 8, h, w, c).transpose(0, 2, 1, 3, 4).reshape(8 * h, 8 * w, c)
        if self.also_save_files:
            from PIL import Image
            os.makedirs(self.log_dir, exist_ok=True)
            img = (grid * 255).astype(np.uint8)
            Image.fromarray(img[..., 0] if c == 1 else img).save(os.path.join(self.log_dir, f"samples_grid_{self.samples_seen:06}.png"))
        return grid


class SaveModelCallback(ExecuteEveryNExamplesCallback):
    """callbacks.py:239-246."""

    def __init__(self, checkpoint_manager, n: int = 10_000):
        super().__init__(n=n)
        self.manager = checkpoint_manager

    def function(self, batch, logs):
        self.manager.save(self.samples_seen)


class LogMetricsCallback(ExecuteEveryNExamplesCallback):
    """callbacks.py:249-268."""

    def __init__(self, every_n_examples: int = 100):
        super().__init__(n=every_n_examples)

    def on_train_begin(self, logs):
        self.samples_seen = self.model.n_img.numpy()

    def function(self, batch: int, logs: Dict):
        self.write_metric_summaries(logs, prefix="batch_")

    def on_epoch_end(self, epoch: int, logs: Dict):
        self.write_metric_summaries(logs, prefix="epoch_")

    def write_metric_summaries(self, logs: Dict, prefix="", flush=False):
        with self.model.summary_writer.as_default() as w:
            for name, value in logs.items():
                if name not in ("batch", "size"):
                    w.scalar(f"{prefix}{name}", value, step=int(self.model.n_img))


class FeedImagesToMetricCallback(ExecuteEveryNExamplesCallback):
    """callbacks.py:138-184: every ``every_n_examples`` start recording ``model.images`` until ``num_samples`` have been
    fed to ``metric.update_state(reals, fakes)``, then write the result and reset."""

    def __init__(self, metric, image_preprocessing_fn, num_samples=1000, every_n_examples=10_000):
        super().__init__(n=every_n_examples, starting_from=-num_samples)
        self.num_samples_per_measurement = num_samples
        self.recording = False
        self.samples_recorded = 0
        self.image_preprocessing_fn = image_preprocessing_fn
        self.metric = metric
        self.results = []

    def function(self, batch, logs):
        self.recording = True

    def on_batch_end(self, batch: int, logs: Dict):
        super().on_batch_end(batch, logs)
        if not self.recording:
            return
        fakes, reals = self.model.images
        n = min(logs["size"], self.num_samples_per_measurement - self.samples_recorded)
        self.metric.update_state(self.image_preprocessing_fn(reals[:n]), self.image_preprocessing_fn(fakes[:n]))
        self.samples_recorded += n
        if self.samples_recorded >= self.num_samples_per_measurement:
            assert self.samples_recorded == self.num_samples_per_measurement
            self.write_result()
            self.recording = False
            self.metric.reset_states()
            self.samples_recorded = 0

    def write_result(self):
        result = self.metric.result()
        self.results.append(result)
        with self.model.summary_writer.as_default() as w:
            w.scalar(self.metric.name, result, step=int(self.model.n_img))


class SWDMetricCallback(FeedImagesToMetricCallback):
    """callbacks.py:186-198 (the reference's ``write_result`` reads an undefined ``self.swd_metric``; fixed)."""

    def __init__(self, image_preprocessing_fn, num_samples=1000, every_n_examples=10_000, seed=None, on_device=False):
        """``on_device``: the preprocessed minibatches (device tensors) are not copied to the host; the metric's device path runs."""
        from .metrics import SWDMetric
        super().__init__(SWDMetric(seed=seed, on_device=on_device), image_preprocessing_fn, num_samples=num_samples,
                         every_n_examples=every_n_examples)

    def write_result(self):
        results = self.metric.results()
        self.results.append(results)
        print(" - " + " - ".join(f"{name}: {value:.4f}" for name, value in results.items()))
        with self.model.summary_writer.as_default() as w:
            for name, value in results.items():
                w.scalar(f"swd/{name}", value, step=int(self.model.n_img))


class FIDMetricCallback(FeedImagesToMetricCallback):
    """callbacks.py:201-206 with the feature extractor injected (the reference downloads Inception-v3 from tfhub.dev)."""

    def __init__(self, image_preprocessing_fn, feature_extractor, num_samples=1000, every_n_examples=10_000):
        from .metrics import FIDMetric
        super().__init__(FIDMetric(feature_extractor), image_preprocessing_fn, num_samples=num_samples, every_n_examples=every_n_examples)
