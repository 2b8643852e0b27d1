"""Tensor-level wrappers over the C ABI (include/bgan.h).  torch is only the container: every call
hands raw device pointers + torch's current HIP stream to libbgan_hip.so.  Nothing here computes
with torch ops, and nothing falls back to the CPU."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, program
from ._lib import Epilogue, EPI_NONE, EPI_BIAS_LRELU, EPI_MUL_GRAD, EPI_TANH, EPI_AFFINE_LRELU, check

LRELU_ALPHA = 0.3


_dev_index = None          # device the last launch's tensors were verified against (refreshed whenever a tensor disagrees)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """Raw handle of torch's CURRENT stream on the CURRENT device (one process per GPU; `_ptr` has checked that every
    operand lives there).  The raw-stream query is a single C call; ``torch.cuda.current_stream()`` builds a Python object
    per call and was a third of the host time of a step."""
    if _raw_stream is None or _get_device is None:
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    cur = _get_device()
    if _dev_index is not None and cur != _dev_index:
        raise BgDeviceError(f"operands live on cuda:{_dev_index} but the current device is cuda:{cur}")
    return _raw_stream(cur)            # a plain int: the bindings declare the parameter as c_void_p and ctypes converts


def _ptr(t):
    """Raw device address of a contiguous tensor that lives on the process's current device.  A CPU tensor or a tensor of
    another GPU raises: the kernels are launched on the current device's stream and must never see foreign pointers."""
    global _dev_index
    if t is None:
        return None
    d = t.get_device()
    if d != _dev_index:
        if d < 0:
            raise BgDeviceError("tensor must live on the GPU (cuda/ROCm device); the HIP path has no CPU fallback")
        cur = torch.cuda.current_device()
        if d != cur:
            raise BgDeviceError(f"tensor lives on cuda:{d} but the current device is cuda:{cur}: call "
                                "torch.cuda.set_device(local_rank) before building the model (one process per GPU)")
        _dev_index = cur
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    if program.active() is not None:
        program.active().keep.append(t)      # a step program holds this address: the tensor lives as long as the program
    return t.data_ptr()                # a plain int (c_void_p parameter): one Python object less per operand and launch


class BgDeviceError(RuntimeError):
    pass


def _f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise ValueError(f"expected float32, got {t.dtype}")


def same_out(n, s):
    return -(-n // s)


def conv2d_splitk_workspace_bytes(bwd_data, B, H, W, Cin, Cout, ksize, stride):
    return _lib.load().bg_conv2d_splitk_workspace_bytes(int(bwd_data), B, H, W, Cin, Cout, ksize, stride)


def epilogue(mode=EPI_NONE, bias=None, ref=None, keep=None, alpha=LRELU_ALPHA, scale=1.0, ws=None, keep_elems=0, stats=None):
    e = Epilogue()
    e.keep_elems = int(keep_elems)
    e.stats = stats.data_ptr() if stats is not None else None
    e.stats_capacity = stats.numel() if stats is not None else 0
    e._rows = C.c_int(0)                    # out: rows of `stats` the conv call wrote (conv2d_stats_rows(e))
    e.stats_rows = C.pointer(e._rows) if stats is not None else None
    e.splitk_ws = ws.data_ptr() if ws is not None else None
    e.splitk_ws_bytes = ws.numel() * ws.element_size() if ws is not None else 0
    e.mode = mode
    e.bias = bias.data_ptr() if bias is not None else None
    e.ref = ref.data_ptr() if ref is not None else None
    e.keep = keep.data_ptr() if keep is not None else None
    e.alpha, e.scale = alpha, scale
    e._hold = (bias, ref, keep, ws, stats)  # keep the tensors alive for the duration of the launch call
    return e


# ------------------------------------------------------------------ blur
def blur_policy(sigma, H, W):
    ks, se, nt = C.c_float(), C.c_float(), C.c_int()
    check(_lib.load().bg_blur_policy(float(sigma), H, W, C.byref(ks), C.byref(se), C.byref(nt)), "bg_blur_policy")
    return ks.value, se.value, nt.value


def gauss_kernel_1d(sigma_eff, kernel_size):
    buf = (C.c_float * 1024)()
    n = C.c_int()
    check(_lib.load().bg_gauss_kernel_1d(float(sigma_eff), float(kernel_size), buf, 1024, C.byref(n)), "bg_gauss_kernel_1d")
    return [buf[i] for i in range(n.value)]


def blur_workspace_bytes(B, H, W, Cc, T):
    return _lib.load().bg_blur_workspace_bytes(B, H, W, Cc, T)


def blur_nhwc(x, y, taps_d, n_taps, tmp=None):
    _f32(x, y, taps_d)
    B, H, W, Cc = x.shape
    assert y.shape == x.shape and taps_d.numel() >= n_taps
    check(_lib.load().bg_blur_nhwc_f32(_ptr(x), _ptr(y), B, H, W, Cc, _ptr(taps_d), n_taps, _ptr(tmp), _stream()), "bg_blur_nhwc_f32")
    return y


def blur3_lerp_supported(B, H, W, Cc, n_taps):
    return bool(_lib.load().bg_blur3_lerp_supported(B, H, W, Cc, n_taps))


def blur3_lerp(f, r, alpha_b, y3, taps_d, n_taps):
    """y3 = [blur(f); blur(r); blur(r + alpha (f - r))] in one launch (include/bgan.h bg_blur3_lerp_nhwc_f32)."""
    _f32(f, r, alpha_b, y3, taps_d)
    B, H, W, Cc = f.shape
    assert r.shape == f.shape and tuple(y3.shape) == (3 * B, H, W, Cc) and alpha_b.numel() == B
    check(_lib.load().bg_blur3_lerp_nhwc_f32(_ptr(f), _ptr(r), _ptr(alpha_b), _ptr(y3), B, H, W, Cc, _ptr(taps_d), n_taps, _stream()),
          "bg_blur3_lerp_nhwc_f32")
    return y3


# ------------------------------------------------------------------ conv family
def conv2d_fwd(x, wT, y, ksize, stride, epi=None):
    """x [B,H,W,Cin], wT [k*k,Cout,Cin] -> y [B,Ho,Wo,Cout]."""
    _f32(x, wT, y)
    B, H, W, Cin = x.shape
    Cout = y.shape[3]
    assert wT.numel() == ksize * ksize * Cin * Cout, (wT.shape, Cin, Cout)
    assert tuple(y.shape) == (B, same_out(H, stride), same_out(W, stride), Cout), (x.shape, y.shape)
    check(_lib.load().bg_conv2d_fwd(_ptr(x), _ptr(wT), _ptr(y), B, H, W, Cin, Cout, ksize, stride,
                                    C.byref(epi) if epi is not None else None, _stream()), "bg_conv2d_fwd")
    return y


def conv2d_bwd_data(dy, w, dx, ksize, stride, epi=None):
    """dy [B,Ho,Wo,Cout], w [k*k,Cin,Cout] -> dx [B,H,W,Cin]."""
    _f32(dy, w, dx)
    B, H, W, Cin = dx.shape
    Cout = dy.shape[3]
    assert w.numel() == ksize * ksize * Cin * Cout
    assert tuple(dy.shape) == (B, same_out(H, stride), same_out(W, stride), Cout), (dy.shape, dx.shape)
    check(_lib.load().bg_conv2d_bwd_data(_ptr(dy), _ptr(w), _ptr(dx), B, H, W, Cin, Cout, ksize, stride,
                                         C.byref(epi) if epi is not None else None, _stream()), "bg_conv2d_bwd_data")
    return dx


def conv2d_bwd_filter_workspace_bytes(B, H, W, Cin, Cout, ksize, stride):
    return _lib.load().bg_conv2d_bwd_filter_workspace_bytes(B, H, W, Cin, Cout, ksize, stride)


def conv2d_bwd_filter(x, dy, dw, ksize, stride, beta=0.0, scale=1.0, ws=None):
    """x [B,H,W,Cin], dy [B,Ho,Wo,Cout] -> dw [k,k,Cin,Cout] = beta*dw + scale*grad."""
    _f32(x, dy, dw)
    B, H, W, Cin = x.shape
    Cout = dy.shape[3]
    assert dw.numel() == ksize * ksize * Cin * Cout
    assert tuple(dy.shape[:3]) == (B, same_out(H, stride), same_out(W, stride))
    wsb = ws.numel() * ws.element_size() if ws is not None else 0
    check(_lib.load().bg_conv2d_bwd_filter(_ptr(x), _ptr(dy), _ptr(dw), B, H, W, Cin, Cout, ksize, stride, beta, scale,
                                           _ptr(ws), wsb, _stream()), "bg_conv2d_bwd_filter")
    return dw


def transpose_last2(src, dst, T, R, Cc):
    _f32(src, dst)
    assert src.numel() == T * R * Cc == dst.numel()
    check(_lib.load().bg_transpose_last2(_ptr(src), _ptr(dst), T, R, Cc, _stream()), "bg_transpose_last2")
    return dst


def transpose_last2_batched(src_base, dst_base, desc, n, total_tiles):
    """desc: int32 device tensor [n, 6] = (src_off, dst_off, T, R, C, first_tile); see include/bgan.h."""
    _f32(src_base, dst_base)
    assert desc.dtype == torch.int32 and desc.is_contiguous() and desc.numel() == 6 * n
    check(_lib.load().bg_transpose_last2_batched(_ptr(src_base), _ptr(dst_base), _ptr(desc), n, total_tiles, _stream()),
          "bg_transpose_last2_batched")
    return dst_base


# ------------------------------------------------------------------ dense / reductions / BN
def gemm(A, Bm, Cm, M, N, K, transA=False, transB=False, bias=None, beta=0.0, scale=1.0):
    _f32(A, Bm, Cm, bias)
    assert A.numel() == M * K and Bm.numel() == K * N and Cm.numel() == M * N
    check(_lib.load().bg_gemm_f32(_ptr(A), _ptr(Bm), _ptr(Cm), M, N, K, int(transA), int(transB), _ptr(bias), beta, scale,
                                  _stream()), "bg_gemm_f32")
    return Cm


def colsum_workspace_bytes(M, N):
    return _lib.load().bg_colsum_workspace_bytes(M, N)


def colsum(x, out, M, N, ws, square=False, beta=0.0, scale=1.0):
    _f32(x, out)
    assert x.numel() == M * N and out.numel() == N
    check(_lib.load().bg_colsum_f32(_ptr(x), _ptr(out), M, N, int(square), beta, scale, _ptr(ws),
                                    ws.numel() * ws.element_size(), _stream()), "bg_colsum_f32")
    return out


def bn_train_fwd(x, y, M, Cc, gamma, beta, moving_mean, moving_var, save_mean, save_inv, ws, eps=1e-3, momentum=0.99,
                 unbiased=True, lrelu_alpha=LRELU_ALPHA):
    _f32(x, y, gamma, beta, save_mean, save_inv)
    assert x.numel() == M * Cc == y.numel()
    check(_lib.load().bg_bn_train_fwd(_ptr(x), _ptr(y), M, Cc, _ptr(gamma), _ptr(beta), _ptr(moving_mean), _ptr(moving_var),
                                      _ptr(save_mean), _ptr(save_inv), eps, momentum, int(unbiased), lrelu_alpha, _ptr(ws),
                                      ws.numel() * ws.element_size(), _stream()), "bg_bn_train_fwd")
    return y


def conv2d_stats_rows(epi):
    """Rows of epilogue(stats=...) that the conv2d_fwd / conv2d_bwd_data call given ``epi`` wrote (0: that geometry took a
    kernel without the statistics epilogue -- run the normal statistics pass).  Returned through the call's own epilogue."""
    return epi._rows.value if epi is not None and epi.stats else 0


def bn_train_fwd_partials(partial, nrows, x, y, M, Cc, gamma, beta, moving_mean, moving_var, save_mean, save_inv, eps=1e-3, momentum=0.99,
                          unbiased=True, lrelu_alpha=LRELU_ALPHA):
    _f32(x, y, partial)
    assert x.numel() == M * Cc == y.numel() and partial.numel() >= nrows * 2 * Cc
    check(_lib.load().bg_bn_train_fwd_partials(_ptr(partial), nrows, _ptr(x), _ptr(y), M, Cc, _ptr(gamma), _ptr(beta), _ptr(moving_mean),
                                               _ptr(moving_var), _ptr(save_mean), _ptr(save_inv), eps, momentum, int(unbiased), lrelu_alpha,
                                               _stream()), "bg_bn_train_fwd_partials")
    return y


def bn_sums_from_partials(partial, nrows, Cc, sums):
    check(_lib.load().bg_bn_sums_from_partials(_ptr(partial), nrows, Cc, _ptr(sums), _stream()), "bg_bn_sums_from_partials")
    return sums


def bn_infer_fwd(x, y, M, Cc, gamma, beta, moving_mean, moving_var, eps=1e-3, lrelu_alpha=LRELU_ALPHA):
    _f32(x, y)
    assert x.numel() == M * Cc == y.numel()
    check(_lib.load().bg_bn_infer_fwd(_ptr(x), _ptr(y), M, Cc, _ptr(gamma), _ptr(beta), _ptr(moving_mean), _ptr(moving_var),
                                      eps, lrelu_alpha, _stream()), "bg_bn_infer_fwd")
    return y


def bn_train_bwd(dy, y, x, dx, M, Cc, gamma, save_mean, save_inv, dgamma, dbeta, ws, lrelu_alpha=LRELU_ALPHA, beta=None):
    """``y`` may be None when ``beta`` is given: the activation's sign is re-derived from ``x`` (bit-identical, one tensor less per
    pass; include/bgan.h at bg_bn_train_bwd).  The same holds for bn_bwd_stats / bn_bwd_apply."""
    _f32(dy, x, dx) if y is None else _f32(dy, y, x, dx)
    assert dy.numel() == M * Cc == dx.numel()
    check(_lib.load().bg_bn_train_bwd(_ptr(dy), _ptr(y), _ptr(x), _ptr(dx), M, Cc, _ptr(gamma), _ptr(beta), _ptr(save_mean), _ptr(save_inv),
                                      _ptr(dgamma), _ptr(dbeta), lrelu_alpha, _ptr(ws), ws.numel() * ws.element_size(),
                                      _stream()), "bg_bn_train_bwd")
    return dx


def bn_fold(gamma, beta, moving_mean, moving_var, eps, scale_out, shift_out):
    check(_lib.load().bg_bn_fold_f32(_ptr(gamma), _ptr(beta), _ptr(moving_mean), _ptr(moving_var), eps, gamma.numel(), _ptr(scale_out),
                                     _ptr(shift_out), _stream()), "bg_bn_fold_f32")


FOLD_MAX = 8


def bn_fold_many(layers):
    """``layers``: up to FOLD_MAX tuples (gamma, beta, moving_mean, moving_var, eps, scale_out, shift_out) folded in ONE launch."""
    import ctypes as C
    n = len(layers)
    assert 0 < n <= FOLD_MAX
    P = C.c_void_p * n
    cols = [P(*[_ptr(l[j]) for l in layers]) for j in (0, 1, 2, 3, 5, 6)]
    eps = (C.c_float * n)(*[float(l[4]) for l in layers])
    cs = (C.c_int * n)(*[int(l[0].numel()) for l in layers])
    check(_lib.load().bg_bn_fold_many_f32(n, cols[0], cols[1], cols[2], cols[3], eps, cs, cols[4], cols[5], _stream()),
          "bg_bn_fold_many_f32")


def bn_stats(x, M, Cc, sums, ws):
    check(_lib.load().bg_bn_stats_f32(_ptr(x), M, Cc, _ptr(sums), _ptr(ws), ws.numel() * ws.element_size(), _stream()), "bg_bn_stats_f32")
    return sums


def bn_finalize(sums, M_total, Cc, save_mean, save_inv, moving_mean, moving_var, eps=1e-3, momentum=0.99, unbiased=True):
    check(_lib.load().bg_bn_finalize_f32(_ptr(sums), M_total, Cc, _ptr(save_mean), _ptr(save_inv), _ptr(moving_mean), _ptr(moving_var),
                                         eps, momentum, int(unbiased), _stream()), "bg_bn_finalize_f32")


def bn_apply(x, y, M, Cc, gamma, beta, mean, inv, lrelu_alpha=LRELU_ALPHA):
    check(_lib.load().bg_bn_apply_f32(_ptr(x), _ptr(y), M, Cc, _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(inv), lrelu_alpha, _stream()),
          "bg_bn_apply_f32")
    return y


def bn_finalize_apply(sums, M_total, x, y, M, Cc, gamma, beta, save_mean, save_inv, moving_mean, moving_var, eps=1e-3, momentum=0.99,
                      unbiased=True, lrelu_alpha=LRELU_ALPHA):
    check(_lib.load().bg_bn_finalize_apply_f32(_ptr(sums), M_total, _ptr(x), _ptr(y), M, Cc, _ptr(gamma), _ptr(beta), _ptr(save_mean),
                                               _ptr(save_inv), _ptr(moving_mean), _ptr(moving_var), eps, momentum, int(unbiased), lrelu_alpha,
                                               _stream()), "bg_bn_finalize_apply_f32")
    return y


def bn_bwd_stats(dy, y, x, M, Cc, save_mean, save_inv, sums, ws, lrelu_alpha=LRELU_ALPHA, gamma=None, beta=None):
    check(_lib.load().bg_bn_bwd_stats_f32(_ptr(dy), _ptr(y), _ptr(x), M, Cc, _ptr(gamma), _ptr(beta), _ptr(save_mean), _ptr(save_inv), lrelu_alpha, _ptr(sums),
                                          _ptr(ws), ws.numel() * ws.element_size(), _stream()), "bg_bn_bwd_stats_f32")
    return sums


def bn_bwd_apply(dy, y, x, dx, M, M_total, Cc, gamma, save_mean, save_inv, sums, lrelu_alpha=LRELU_ALPHA, beta=None):
    check(_lib.load().bg_bn_bwd_apply_f32(_ptr(dy), _ptr(y), _ptr(x), _ptr(dx), M, M_total, Cc, _ptr(gamma), _ptr(beta), _ptr(save_mean),
                                          _ptr(save_inv), _ptr(sums), lrelu_alpha, _stream()), "bg_bn_bwd_apply_f32")
    return dx


def bn_param_grads(sums, Cc, scale, dgamma, dbeta):
    check(_lib.load().bg_bn_param_grads_f32(_ptr(sums), Cc, scale, _ptr(dgamma), _ptr(dbeta), _stream()), "bg_bn_param_grads_f32")


# ------------------------------------------------------------------ pointwise / losses / adam / rng
def lerp(r, f, alpha_b, out):
    B = r.shape[0]
    n_per = r.numel() // B
    check(_lib.load().bg_lerp_f32(_ptr(r), _ptr(f), _ptr(alpha_b), _ptr(out), B, n_per, _stream()), "bg_lerp_f32")
    return out


def row_norm(g, out_b):
    B = g.shape[0]
    check(_lib.load().bg_row_norm_f32(_ptr(g), _ptr(out_b), B, g.numel() // B, _stream()), "bg_row_norm_f32")
    return out_b


def gp_seed(g, norm_b, coef, out, zero_norm_guard=False):
    B = g.shape[0]
    fn = _lib.load().bg_gp_seed_guarded_f32 if zero_norm_guard else _lib.load().bg_gp_seed_f32
    check(fn(_ptr(g), _ptr(norm_b), coef, _ptr(out), B, g.numel() // B, _stream()), "bg_gp_seed_f32")
    return out


def mul_grad(d, ref, out, keep=None, alpha=LRELU_ALPHA, scale=1.0):
    assert d.numel() == ref.numel() == out.numel()
    check(_lib.load().bg_mul_grad_f32(_ptr(d), _ptr(ref), _ptr(keep), alpha, scale, _ptr(out), d.numel(), _stream()), "bg_mul_grad_f32")
    return out


def tanh_bwd(dy, y, out):
    check(_lib.load().bg_tanh_bwd_f32(_ptr(dy), _ptr(y), _ptr(out), dy.numel(), _stream()), "bg_tanh_bwd_f32")
    return out


def outer(s_b, w_k, out):
    B, K = s_b.numel(), w_k.numel()
    assert out.numel() == B * K
    check(_lib.load().bg_outer_f32(_ptr(s_b), _ptr(w_k), _ptr(out), B, K, _stream()), "bg_outer_f32")
    return out


def fill(x, v):
    check(_lib.load().bg_fill_f32(_ptr(x), float(v), x.numel(), _stream()), "bg_fill_f32")
    return x


def scale_(x, v):
    check(_lib.load().bg_scale_f32(_ptr(x), float(v), x.numel(), _stream()), "bg_scale_f32")
    return x


def copy_(dst, src):
    """dst <- src through a kernel of the library (recordable into a step program)."""
    _f32(dst, src)
    assert dst.numel() == src.numel()
    check(_lib.load().bg_copy_f32(_ptr(dst), _ptr(src), dst.numel(), _stream()), "bg_copy_f32")
    return dst


def wgangp_d_loss(fs, rs, norm_b, inv_gbs, gp_coef, e_drift, vec_scale, dfs, drs, metrics):
    B = fs.numel()
    check(_lib.load().bg_wgangp_d_loss(_ptr(fs), _ptr(rs), _ptr(norm_b), B, inv_gbs, gp_coef, e_drift, vec_scale, _ptr(dfs),
                                       _ptr(drs), _ptr(metrics), _stream()), "bg_wgangp_d_loss")


def wgan_g_loss(s, inv_gbs, ds, metrics):
    check(_lib.load().bg_wgan_g_loss(_ptr(s), s.numel(), inv_gbs, _ptr(ds), _ptr(metrics), _stream()), "bg_wgan_g_loss")


def u8_normalize_resize(src_u8, dst):
    """uint8 NHWC -> float32 NHWC in [-1, 1], bilinear-resized to dst's spatial size (demo_celeba.py:22-35)."""
    assert src_u8.dtype == torch.uint8 and dst.dtype == torch.float32
    B, Hs, Ws, Cc = src_u8.shape
    assert dst.shape[0] == B and dst.shape[3] == Cc
    check(_lib.load().bg_u8_normalize_resize_f32(_ptr(src_u8), _ptr(dst), B, Hs, Ws, Cc, dst.shape[1], dst.shape[2], _stream()),
          "bg_u8_normalize_resize_f32")
    return dst


def adam(theta, m, v, g, lr_t, b1=0.9, b2=0.999, eps=1e-7):
    assert theta.numel() == m.numel() == v.numel() == g.numel()
    check(_lib.load().bg_adam_f32(_ptr(theta), _ptr(m), _ptr(v), _ptr(g), theta.numel(), lr_t, b1, b2, eps, _stream()), "bg_adam_f32")


def _draw_offset(out, offset, counter):
    """Offset of a counter-based draw.  ``counter`` = (obj, attr): the draw starts at that host counter, which then advances by the
    Philox blocks the draw consumes; while a step program is being recorded the launch's offset is bound to the counter."""
    if counter is None:
        return offset
    obj, attr = counter
    inc = (out.numel() + 3) // 4
    offset = getattr(obj, attr)
    if program.active() is not None:
        program.active().bind_rng(obj, attr, inc)        # the recorded launch re-reads the counter before every replay
    setattr(obj, attr, offset + inc)
    return offset


def uniform(out, seed, offset=0, counter=None):
    offset = _draw_offset(out, offset, counter)
    check(_lib.load().bg_uniform_f32(_ptr(out), out.numel(), seed, offset, _stream()), "bg_uniform_f32")
    return out


def keep_mask(out, keep_prob, seed, offset=0, counter=None):
    assert out.dtype == torch.uint8
    offset = _draw_offset(out, offset, counter)
    check(_lib.load().bg_keep_mask_u8(_ptr(out), out.numel(), keep_prob, seed, offset, _stream()), "bg_keep_mask_u8")
    return out


# ------------------------------------------------------------------ profiling
_ranges_on = None


class trace_range:
    """Named range on the rocprofv3 marker timeline (roctx through the C ABI).  Off unless BGAN_ROCTX=1: then the first use binds
    roctx; with the switch off (the default) entering and leaving cost one attribute test."""
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global _ranges_on
        if _ranges_on is None:
            import os
            _ranges_on = bool(os.environ.get("BGAN_ROCTX") == "1" and _lib.load().bg_range_enable(1))
        if _ranges_on:
            _lib.load().bg_range_push(self.name.encode())
        return self

    def __exit__(self, *exc):
        if _ranges_on:
            _lib.load().bg_range_pop()
        return False


def prof_enable(on):
    _lib.load().bg_prof_enable(int(on))


def prof_reset():
    _lib.load().bg_prof_reset()


def conv2d_useful_flops(B, H, W, Cin, Cout, k=5, stride=2):
    """Flops of a k x k SAME convolution that multiply real data (no zero-padding taps): bg_conv2d_useful_flops."""
    return float(_lib.load().bg_conv2d_useful_flops(B, H, W, Cin, Cout, k, stride))


def prof_records(with_exec=False, with_useful=False):
    """[(name, ms, algorithmic flops, algorithmic bytes)] per recorded launch; with_exec appends the flops the launch issued
    on the matrix pipe (bg_prof_get_exec), with_useful the flops that multiply real data (bg_prof_get_useful)."""
    lib = _lib.load()
    n = lib.bg_prof_count()
    out = []
    name = C.create_string_buffer(128)
    ms, fl, by, ex = C.c_float(), C.c_double(), C.c_double(), C.c_double()
    for i in range(n):
        check(lib.bg_prof_get(i, name, 128, C.byref(ms), C.byref(fl), C.byref(by)), "bg_prof_get")
        rec = (name.value.decode(), ms.value, fl.value, by.value)
        if with_exec:
            check(lib.bg_prof_get_exec(i, C.byref(ex)), "bg_prof_get_exec")
            rec = rec + (ex.value,)
        if with_useful:
            check(lib.bg_prof_get_useful(i, C.byref(ex)), "bg_prof_get_useful")
            rec = rec + (ex.value,)
        out.append(rec)
    return out
