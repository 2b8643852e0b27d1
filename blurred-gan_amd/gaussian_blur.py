"""Mirror of reference gaussian_blur.py: same public names, HIP kernels underneath.

``GaussianBlur2D`` is the non-trainable layer the reference prepends to the critic
(blurred_gan.py:30-34); ``std`` is assignable exactly like the reference's ``tf.Variable``
(callbacks.py:62: ``model.std.assign(value)``)."""
from __future__ import annotations

import torch

from . import ops
from .layers import Layer, default_device


def maximum_reasonable_std(image_resolution: int) -> float:
    """gaussian_blur.py:15-18."""
    return appropriate_std(image_resolution - 1)


def appropriate_kernel_size(std: float):
    """gaussian_blur.py:21-26 (float result, like the reference's expression)."""
    return (6 * std) * 2 // 2 + 1


def appropriate_std(kernel_size):
    """gaussian_blur.py:29-31."""
    return (kernel_size - 1.0) / 6.0


def get_data_format(image) -> str:
    """gaussian_blur.py:34-39."""
    return "NHWC" if image.shape[-1] in (1, 3) else "NCHW"


def get_image_dims(image):
    """gaussian_blur.py:42-47."""
    fmt = get_data_format(image)
    h = image.shape[1 if fmt == "NHWC" else 2]
    w = image.shape[2 if fmt == "NHWC" else -1]
    c = image.shape[-1 if fmt == "NHWC" else 1]
    return h, w, c


def gaussian_kernel_1d(std, kernel_size):
    """gaussian_blur.py:83-88 -> float32 tensor of 2*floor(ks/2)+1 taps (host maths in the C ABI)."""
    return torch.tensor(ops.gauss_kernel_1d(float(std), float(kernel_size)), dtype=torch.float32)


def _as_nhwc(image):
    """The reference picks NCHW when the last dim is not 1 or 3 (gaussian_blur.py:34-39); the kernels are
    NHWC, so NCHW input is permuted (host plumbing) and permuted back."""
    if get_data_format(image) == "NHWC":
        return image.contiguous(), False
    return image.permute(0, 2, 3, 1).contiguous(), True


def gaussian_blur(image, std: float, kernel_size):
    """gaussian_blur.py:91-132."""
    x, was_nchw = _as_nhwc(image.to(default_device(), torch.float32))
    B, H, W, C = x.shape
    taps = gaussian_kernel_1d(std, kernel_size).to(x.device)
    nb = ops.blur_workspace_bytes(B, H, W, C, taps.numel())
    tmp = torch.empty(nb // 4 + 4, dtype=torch.float32, device=x.device) if nb else None
    y = ops.blur_nhwc(x, torch.empty_like(x), taps, taps.numel(), tmp)
    return y.permute(0, 3, 1, 2).contiguous() if was_nchw else y


def blur_images(images, scale: float):
    """gaussian_blur.py:50-80: sigma policy (clip kernel size to [3, max(h,w)], re-derive sigma) + blur."""
    h, w, _ = get_image_dims(images)
    ks, std, _ = ops.blur_policy(float(scale), int(h), int(w))
    return gaussian_blur(images, std, ks)


class Variable:
    """Minimal scalar stand-in for the reference's ``tf.Variable`` (assign / numpy / float)."""

    def __init__(self, value, name=None, dtype=float):
        self._dtype = dtype
        self._v = dtype(value)
        self.name = name

    def assign(self, value):
        self._v = self._dtype(value.numpy() if hasattr(value, "numpy") else value)
        return self

    def assign_add(self, value):
        self._v = self._dtype(self._v + value)
        return self

    def numpy(self):
        return self._v

    def __float__(self):
        return float(self._v)

    def __int__(self):
        return int(self._v)

    def __index__(self):
        return int(self._v)

    def __mod__(self, o):
        return self._v % o

    def __floordiv__(self, o):
        return self._v // o

    def __truediv__(self, o):
        return self._v / o

    def __repr__(self):
        return f"<Variable {self.name}={self._v}>"


class GaussianBlur2D(Layer):
    """gaussian_blur.py:135-148."""
    kind = "blur"

    def __init__(self, initial_std=0.01, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.std = Variable(initial_std, name="std")
        self.trainable = False

    def call(self, image):
        return blur_images(image, float(self.std))

    __call__ = call
