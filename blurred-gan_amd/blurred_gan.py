"""Mirror of reference blurred_gan.py: ``BlurredVariant`` class factory, ``BlurredWGANGP``, ``BlurredWGAN``."""
from __future__ import annotations

from dataclasses import dataclass

from .gaussian_blur import GaussianBlur2D
from .layers import Sequential
from .wgan import WGAN, WGANGP, Mean, TrainingConfig  # noqa: F401  (re-exported like the reference module)


def BlurredVariant(some_gan_base_class):
    class BlurredGAN(some_gan_base_class):
        """Any GAN base class with a non-trainable Gaussian blur prepended to the critic
        (blurred_gan.py:17-49)."""

        @dataclass
        class HyperParameters(some_gan_base_class.HyperParameters):
            initial_blur_std: float = 0.05

        def __init__(self, generator, discriminator, hyperparams, config, **kwargs):
            blur = GaussianBlur2D(initial_std=hyperparams.initial_blur_std, input_shape=discriminator.input_shape[1:])
            discriminator_with_blur = Sequential([blur, discriminator])          # blurred_gan.py:31-34
            super().__init__(generator, discriminator_with_blur, hyperparams=hyperparams, config=config, **kwargs)
            self.blur = blur
            self.std_metric = Mean("std")

        @property
        def std(self):
            return self.blur.std

        @property
        def metrics(self):
            return super().metrics + [self.std_metric]

        def discriminator_step(self, reals):
            disc_loss, images = super().discriminator_step(reals)
            self._after_d_step()
            return disc_loss, images

        def _after_d_step(self):
            """Host side of the step (also run when train_on_batch replays the recorded step program)."""
            self.std_metric(float(self.std))                                     # blurred_gan.py:47

    BlurredGAN.__name__ = BlurredGAN.__qualname__ = "Blurred" + some_gan_base_class.__name__
    return BlurredGAN


BlurredWGANGP = BlurredVariant(WGANGP)
BlurredWGAN = BlurredVariant(WGAN)
