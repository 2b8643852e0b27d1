"""The DCGAN stacks of the reference demos as ``layers.Sequential`` subclasses (same class names).

``arch``: "mnist" (demo_mnist.py:48-86), "celeba128" (demo_celeba.py:51-124, verbatim), "celeba64"
(build-side definition for BASELINE.json's 64x64 configs: the 128 stack minus its outermost stage on each
side, SURVEY.md 8a "Architecture note"), plus two small test geometries ("tiny": MFMA-tileable channel
counts on 8x8x3 images; "tiny_mnist": odd channel counts, 12x12x1, ConvT with fused tanh)."""
from __future__ import annotations

from . import layers

_G = {  # base_hw, dense_ch, [(filters, stride, activation)], last_conv_filters
    "mnist": (7, 256, [(128, 1, None), (64, 2, None), (1, 2, "tanh")], None),
    "celeba128": (4, 512, [(512, 1, None), (256, 2, None), (128, 2, None), (64, 2, None), (32, 2, None), (16, 2, None)], 3),
    "celeba64": (4, 512, [(512, 1, None), (256, 2, None), (128, 2, None), (64, 2, None), (32, 2, None)], 3),
    "tiny": (2, 32, [(32, 1, None), (16, 2, None), (16, 2, None)], 3),
    "tiny_mnist": (3, 8, [(8, 1, None), (4, 2, None), (1, 2, "tanh")], None),
}
_D = {"mnist": [64, 128], "celeba128": [16, 32, 64, 128, 256, 512], "celeba64": [32, 64, 128, 256, 512],
      "tiny": [16, 32], "tiny_mnist": [4, 8]}
IMAGE_SHAPE = {"mnist": (28, 28, 1), "celeba128": (128, 128, 3), "celeba64": (64, 64, 3), "tiny": (8, 8, 3),
               "tiny_mnist": (12, 12, 1)}
LATENT = {"mnist": 100, "celeba128": 100, "celeba64": 100, "tiny": 10, "tiny_mnist": 6}


class DCGANGenerator(layers.Sequential):
    def __init__(self, latent_size=None, arch="celeba128", *args, **kwargs):
        super().__init__(*args, **kwargs)
        base, ch, convt, last = _G[arch]
        self.latent_size = latent_size or LATENT[arch]
        self.add(layers.Dense(base * base * ch, use_bias=False, input_shape=(self.latent_size,)))
        self.add(layers.BatchNormalization())
        self.add(layers.LeakyReLU())
        self.add(layers.Reshape((base, base, ch)))
        assert self.output_shape == (None, base, base, ch)
        hw = base
        for filters, stride, act in convt:
            self.add(layers.Conv2DTranspose(filters, (5, 5), strides=(stride, stride), padding="same", use_bias=False, activation=act))
            hw *= stride
            assert self.output_shape == (None, hw, hw, filters), self.output_shape
            if act is None:
                self.add(layers.BatchNormalization())
                self.add(layers.LeakyReLU())
        if last is not None:
            self.add(layers.Conv2D(last, (5, 5), padding="same", use_bias=False, activation="tanh"))
        assert self.output_shape == (None,) + IMAGE_SHAPE[arch], self.output_shape


class DCGANDiscriminator(layers.Sequential):
    def __init__(self, arch="celeba128", *args, **kwargs):
        super().__init__(*args, **kwargs)
        for i, c in enumerate(_D[arch]):
            kw = dict(input_shape=list(IMAGE_SHAPE[arch])) if i == 0 else {}
            self.add(layers.Conv2D(c, 5, strides=2, padding="same", **kw))
            self.add(layers.LeakyReLU())
            self.add(layers.Dropout(0.3))
        self.add(layers.Flatten())
        self.add(layers.Dense(1, activation="linear"))
