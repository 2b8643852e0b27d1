"""blurred-GAN WGAN-GP training step, MI355X-native (hand-written HIP kernels behind a C ABI).

Import as ``blurred_gan_amd`` (the repo-root shim maps that name onto this directory, whose own name
is not a valid Python identifier).  Module names mirror the reference repository: ``wgan``,
``blurred_gan``, ``gaussian_blur``, ``callbacks``, ``utils``; ``layers`` stands in for
``tensorflow.keras.layers``."""
import os as _os

# Multi-process GPU work on this platform needs the dmabuf IPC mode (RCCL and cross-process tensor sharing fail with
# hipIpcGetMemHandle: invalid argument otherwise); the variable is read when the HSA runtime starts, i.e. at the first HIP call,
# so it is set here, before anything of the package touches the device -- and only if the launcher has not said otherwise.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from . import layers, utils, dist, gaussian_blur, wgan, blurred_gan, callbacks, models, checkpoint, metrics, sliced_wasserstein  # noqa: F401
from .layers import set_seed, Sequential  # noqa: F401
from .gaussian_blur import GaussianBlur2D, blur_images  # noqa: F401
from .wgan import WGAN, WGANGP, TrainingConfig, gradient_penalty  # noqa: F401
from .blurred_gan import BlurredVariant, BlurredWGANGP, BlurredWGAN  # noqa: F401

__all__ = ["layers", "utils", "dist", "gaussian_blur", "wgan", "blurred_gan", "callbacks", "set_seed", "Sequential",
           "GaussianBlur2D", "blur_images", "WGAN", "WGANGP", "TrainingConfig", "gradient_penalty", "BlurredVariant",
           "BlurredWGANGP", "BlurredWGAN"]
