"""Explicit numpy forward/backward formulas for every op on the hot path.

Oracle = test infrastructure (see ``oracle/__init__.py``); parity unpinned.

Every function cites the reference line (``/root/reference``) whose behaviour it
restates, or the TensorFlow op semantics ([TF]) the reference relies on.  All
tensors are NHWC.  ``dtype`` is whatever the inputs carry: tests run the oracle
in float64 (the "true" answer) and in float32 (same rounding class as the HIP
kernels).
"""
from __future__ import annotations

import math
import numpy as np

LRELU_ALPHA = 0.3      # keras.layers.LeakyReLU() default, demo_celeba.py:57 [TF]
BN_MOMENTUM = 0.99     # keras.layers.BatchNormalization() defaults [TF]
BN_EPS = 1e-3
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-7   # tf.keras.optimizers.Adam defaults [TF]


# ----------------------------------------------------------------------------
# Gaussian blur policy + kernel (gaussian_blur.py:15-88)
# ----------------------------------------------------------------------------
def appropriate_kernel_size(std):
    """gaussian_blur.py:21-26 -- ``(6*std)*2//2+1`` evaluated in float32."""
    s = np.float32(std)
    return np.float32(np.floor((np.float32(6) * s) * np.float32(2) / np.float32(2)) + np.float32(1))


def appropriate_std(kernel_size):
    """gaussian_blur.py:29-31."""
    return np.float32((np.float32(kernel_size) - np.float32(1.0)) / np.float32(6.0))


def maximum_reasonable_std(image_resolution):
    """gaussian_blur.py:15-18."""
    return appropriate_std(image_resolution - 1)


def blur_policy(std, h, w):
    """gaussian_blur.py:58-72: sigma -> (kernel_size, sigma_eff, n_taps), all float32 maths.

    kernel_size is clipped to [3, max(h, w)]; sigma is ALWAYS re-derived from the
    kernel size; the tap count is ``2*floor(ks/2)+1`` (gaussian_blur.py:84).
    """
    full = np.float32(max(h, w))
    ks = appropriate_kernel_size(std)
    ks = np.float32(min(max(ks, np.float32(3)), full))
    s = appropriate_std(ks)
    s = np.float32(max(s, np.float32(0.01)))
    half = int(math.floor(float(ks) / 2.0))
    return float(ks), float(s), 2 * half + 1


def gaussian_kernel_1d(std, kernel_size, dtype=np.float32):
    """gaussian_blur.py:83-88 (float32 in the reference; dtype selectable here)."""
    half = int(math.floor(float(kernel_size) / 2.0))
    x = np.arange(-half, half + 1).astype(dtype)
    std = dtype(std)
    g = np.exp(-(x ** 2 / (dtype(2) * std ** 2))) / (np.sqrt(dtype(2 * math.pi)) * std)
    g = g / g.sum(dtype=dtype)
    return g.astype(dtype)


def blur_1d_axis(x, g, axis):
    """[TF] depthwise_conv2d, stride 1, SAME, one spatial axis, zero padding (gaussian_blur.py:116-130)."""
    t = g.shape[0]
    half = t // 2
    n = x.shape[axis]
    pad = [(0, 0)] * x.ndim
    pad[axis] = (half, half)
    xp = np.pad(x, pad)
    out = np.zeros_like(x)
    for j in range(t):
        sl = [slice(None)] * x.ndim
        sl[axis] = slice(j, j + n)
        out = out + g[j] * xp[tuple(sl)]
    return out


def gaussian_blur(x, std, kernel_size):
    """gaussian_blur.py:91-132: pass 1 along H, pass 2 along W, NHWC."""
    g = gaussian_kernel_1d(std, kernel_size, dtype=x.dtype.type)
    return blur_1d_axis(blur_1d_axis(x, g, 1), g, 2)


def blur_images(x, scale):
    """gaussian_blur.py:50-80."""
    ks, s, _ = blur_policy(scale, x.shape[1], x.shape[2])
    return gaussian_blur(x, s, ks)


# ----------------------------------------------------------------------------
# [TF] SAME geometry
# ----------------------------------------------------------------------------
def same_pads(n, k, s):
    """[TF] SAME: out = ceil(n/s); pad_total = max((out-1)*s+k-n, 0); before = total//2."""
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return out, tot // 2, tot - tot // 2


# ----------------------------------------------------------------------------
# Conv2D / Conv2DTranspose (demo_celeba.py:62-119) -- [TF] cross-correlation
# ----------------------------------------------------------------------------
def conv2d_fwd(x, w, stride):
    """y[b,oh,ow,co] = sum x[b, oh*s+kh-pt, ow*s+kw-pl, ci] * w[kh,kw,ci,co]."""
    B, H, W, Ci = x.shape
    kh_, kw_, _, Co = w.shape
    Ho, pt, pb = same_pads(H, kh_, stride)
    Wo, pl, pr = same_pads(W, kw_, stride)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    y = np.zeros((B, Ho, Wo, Co), dtype=x.dtype)
    for kh in range(kh_):
        for kw in range(kw_):
            xs = xp[:, kh:kh + stride * Ho:stride, kw:kw + stride * Wo:stride, :]
            y += xs @ w[kh, kw]
    return y


def conv2d_bwd_data(dy, w, stride, in_hw):
    """Gradient of conv2d_fwd w.r.t. x; also Conv2DTranspose forward ([TF] conv2d_backprop_input)."""
    B, Ho, Wo, Co = dy.shape
    kh_, kw_, Ci, _ = w.shape
    H, W = in_hw
    Ho2, pt, pb = same_pads(H, kh_, stride)
    Wo2, pl, pr = same_pads(W, kw_, stride)
    assert (Ho2, Wo2) == (Ho, Wo), ((Ho2, Wo2), (Ho, Wo))
    dxp = np.zeros((B, H + pt + pb, W + pl + pr, Ci), dtype=dy.dtype)
    for kh in range(kh_):
        for kw in range(kw_):
            dxp[:, kh:kh + stride * Ho:stride, kw:kw + stride * Wo:stride, :] += dy @ w[kh, kw].T
    return dxp[:, pt:pt + H, pl:pl + W, :]


def conv2d_bwd_filter(x, dy, stride, ksize):
    """Gradient of conv2d_fwd w.r.t. w -> [kh,kw,ci,co]."""
    B, H, W, Ci = x.shape
    _, Ho, Wo, Co = dy.shape
    _, pt, pb = same_pads(H, ksize, stride)
    _, pl, pr = same_pads(W, ksize, stride)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    dw = np.zeros((ksize, ksize, Ci, Co), dtype=x.dtype)
    dy2 = dy.reshape(-1, Co)
    for kh in range(ksize):
        for kw in range(ksize):
            xs = xp[:, kh:kh + stride * Ho:stride, kw:kw + stride * Wo:stride, :]
            dw[kh, kw] = xs.reshape(-1, Ci).T @ dy2
    return dw


def conv2d_transpose_fwd(x, w, stride):
    """Keras Conv2DTranspose(padding='same'): kernel [kh,kw,c_out,c_in], output H*s ([TF])."""
    B, h, w_, Ci = x.shape
    return conv2d_bwd_data(x, w, stride, (h * stride, w_ * stride))


def conv2d_transpose_bwd_data(dy, w, stride):
    return conv2d_fwd(dy, w, stride)


def conv2d_transpose_bwd_filter(x, dy, stride, ksize):
    """dW of Conv2DTranspose: roles of x/dy swap relative to the underlying conv."""
    return conv2d_bwd_filter(dy, x, stride, ksize)


# ----------------------------------------------------------------------------
# Dense, BN, activations, dropout
# ----------------------------------------------------------------------------
def dense_fwd(x, w, b=None):
    y = x @ w
    return y if b is None else y + b


def dense_bwd(x, w, dy, need_dx=True):
    dw = x.T @ dy
    db = dy.sum(0)
    dx = dy @ w.T if need_dx else None
    return dx, dw, db


def _bn_axes(x):
    return tuple(range(x.ndim - 1))


def bn_train_fwd(x, gamma, beta, mov_mean, mov_var):
    """[TF] BatchNormalization(training=True): biased batch variance normalises; the moving
    variance receives the UNBIASED variance for 4-D inputs (fused kernel) and the biased one
    for 2-D inputs."""
    ax = _bn_axes(x)
    n = int(np.prod([x.shape[a] for a in ax]))
    mean = x.mean(ax)
    var = ((x - mean) ** 2).mean(ax)
    inv = 1.0 / np.sqrt(var + x.dtype.type(BN_EPS))
    xhat = (x - mean) * inv
    y = gamma * xhat + beta
    var_upd = var * (n / max(n - 1, 1)) if x.ndim == 4 else var
    mom = x.dtype.type(BN_MOMENTUM)
    new_mean = mov_mean * mom + mean * (1 - mom)
    new_var = mov_var * mom + var_upd * (1 - mom)
    return y, (xhat, inv), new_mean.astype(x.dtype), new_var.astype(x.dtype)


def bn_infer_fwd(x, gamma, beta, mov_mean, mov_var):
    inv = 1.0 / np.sqrt(mov_var + x.dtype.type(BN_EPS))
    return gamma * (x - mov_mean) * inv + beta


def bn_train_bwd(dy, gamma, cache):
    xhat, inv = cache
    ax = _bn_axes(dy)
    n = int(np.prod([dy.shape[a] for a in ax]))
    dbeta = dy.sum(ax)
    dgamma = (dy * xhat).sum(ax)
    dx = (gamma * inv / n) * (n * dy - dbeta - xhat * dgamma)
    return dx, dgamma, dbeta


def lrelu_fwd(x, alpha=LRELU_ALPHA):
    return np.where(x > 0, x, x.dtype.type(alpha) * x)


def lrelu_mask(x, alpha=LRELU_ALPHA):
    """[TF] LeakyReluGrad: features > 0 ? 1 : alpha."""
    return np.where(x > 0, x.dtype.type(1), x.dtype.type(alpha))


def dropout_fwd(x, keep_mask, rate):
    """[TF] Dropout(training=True) with an explicit keep mask (1 = kept)."""
    scale = x.dtype.type(1.0 / (1.0 - rate))
    return x * scale * keep_mask.astype(x.dtype)


# ----------------------------------------------------------------------------
# Adam ([TF] keras OptimizerV2 Adam)
# ----------------------------------------------------------------------------
def adam_update(theta, m, v, g, t, lr):
    dt = theta.dtype.type
    b1, b2 = dt(ADAM_B1), dt(ADAM_B2)
    lr_t = dt(lr * math.sqrt(1.0 - ADAM_B2 ** t) / (1.0 - ADAM_B1 ** t))
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    theta = theta - lr_t * m / (np.sqrt(v) + dt(ADAM_EPS))
    return theta, m, v


def exponential_decay(initial, step, decay_steps, rate):
    """[TF] ExponentialDecay(staircase=False) as used by callbacks.py:51-57, float32."""
    p = np.float32(step) / np.float32(decay_steps)
    return float(np.float32(initial) * np.float32(np.power(np.float32(rate), p)))


# ----------------------------------------------------------------------------
# Input pipeline (demo_celeba.py:22-35): normalise, then [TF] tf.image.resize bilinear (half-pixel centres)
# ----------------------------------------------------------------------------
def normalize_resize_bilinear(img_u8, out_hw, dtype=np.float64):
    """uint8 NHWC -> (x - 127.5) / 127.5 -> bilinear resize to out_hw.  [TF] resize_bilinear(half_pixel_centers=True):
    in = (out + 0.5) * scale - 0.5, lower = max(floor(in), 0), upper = min(ceil(in), size - 1), lerp = in - floor(in)."""
    x = (img_u8.astype(dtype) - dtype(127.5)) / dtype(127.5)
    B, Hs, Ws, C = x.shape
    Hd, Wd = out_hw

    def weights(n_out, n_in):
        pos = (np.arange(n_out, dtype=dtype) + dtype(0.5)) * dtype(n_in / n_out) - dtype(0.5)
        lo_f = np.floor(pos)
        lo = np.maximum(lo_f, 0).astype(int)
        hi = np.minimum(np.ceil(pos), n_in - 1).astype(int)
        return lo, hi, (pos - lo_f).astype(dtype)
    y0, y1, ly = weights(Hd, Hs)
    x0, x1, lx = weights(Wd, Ws)
    top = x[:, y0][:, :, x0] + (x[:, y0][:, :, x1] - x[:, y0][:, :, x0]) * lx[None, None, :, None]
    bot = x[:, y1][:, :, x0] + (x[:, y1][:, :, x1] - x[:, y1][:, :, x0]) * lx[None, None, :, None]
    return top + (bot - top) * ly[None, :, None, None]
