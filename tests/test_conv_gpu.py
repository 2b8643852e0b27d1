"""GPU parity: conv forward / data-gradient / filter-gradient kernels (MFMA, thin and direct paths) vs the
float64 oracle, plus adjointness at full BASELINE sizes."""
import numpy as np
import pytest
import torch

from oracle import np_ops as O
from helpers import dev, conv_tol

pytestmark = pytest.mark.gpu

# (B, H, W, Cin, Cout, stride) -- covers: MFMA BK=32/16, N tails (16), thin-K (Cin 3/1), thin-N (Cout 3/1),
# direct fallback (8->4), odd spatial sizes, stride 1 and 2, M not a multiple of the tile
CASES = [
    (2, 8, 8, 32, 32, 2), (3, 8, 8, 32, 64, 1), (2, 16, 16, 64, 128, 2), (2, 4, 4, 128, 256, 2), (5, 2, 2, 256, 512, 2),
    (2, 8, 8, 16, 32, 2), (2, 8, 8, 32, 16, 2), (2, 16, 16, 16, 16, 1), (3, 7, 7, 32, 32, 2), (2, 7, 9, 32, 64, 1),
    (2, 16, 16, 3, 32, 2), (2, 28, 28, 1, 64, 2), (2, 16, 16, 32, 3, 1), (2, 14, 14, 64, 1, 2), (2, 12, 12, 8, 4, 2),
    (2, 6, 6, 4, 8, 1), (1, 32, 32, 64, 32, 2), (9, 4, 4, 512, 512, 1),
    (3, 20, 12, 16, 3, 1), (2, 40, 16, 32, 2, 1),          # thin-Co row kernels: Ci 16 / 32, ragged row blocks
    (3, 24, 24, 1, 64, 2), (2, 18, 16, 3, 16, 1), (2, 36, 32, 2, 32, 2),   # thin-Ci row kernels: Co 64 / 16 / 32, ragged row blocks
    (3, 32, 32, 3, 32, 2), (3, 64, 64, 3, 16, 2), (5, 24, 32, 16, 3, 1), (2, 20, 16, 3, 32, 1), (3, 36, 64, 64, 2, 1),   # scatter-form row kernels (thin-N fwd / dgrad)
    (3, 27, 25, 1, 64, 2), (2, 44, 40, 2, 16, 2), (5, 13, 14, 4, 32, 2),         # data gradients to 1 / 2 / 4 channels: all phases from one staged patch (odd sizes: phases of different extents; several tiles)
    (2, 20, 128, 16, 3, 1), (2, 24, 256, 3, 16, 2),                           # ... on 128-pixel rows (8 waves per workgroup)
    (128, 4, 4, 64, 64, 1), (128, 8, 8, 32, 64, 2), (128, 8, 8, 64, 32, 2),   # position-major tiles: padding taps skipped (64- and 128-row tiles)
    (1024, 16, 16, 16, 32, 2),                                                 # position-major AND the four phases merged in one workgroup
    (384, 4, 4, 128, 256, 2), (384, 8, 8, 64, 128, 2),                         # position-major filter gradient at a batch that is whole 128-image chunks but no power of two (the merged critic pass: 3 x 128)
    (3, 64, 64, 16, 32, 2), (2, 128, 128, 16, 32, 2), (5, 12, 128, 16, 32, 2), (2, 4, 64, 16, 32, 2),   # row-staged 16-channel kernels (C4's outer layers)
    # strip-resident filter gradient (Ci % 32 == 0, Co % 64 == 0, stride 2, Wo = 8 / 16 / 32): one strip per image, several strips per
    # image, non-square maps, a strip count the workgroups do not divide, several channel tiles
    (3, 16, 16, 32, 64, 2), (2, 32, 32, 64, 128, 2), (2, 64, 64, 32, 64, 2), (5, 32, 16, 32, 64, 2), (7, 16, 16, 64, 64, 2), (2, 8, 64, 96, 192, 2),
]


def _data(B, H, W, Ci, Co, s, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, size=(B, H, W, Ci))
    w = rng.uniform(-1, 1, size=(5, 5, Ci, Co)) / np.sqrt(25 * Ci)
    Ho, Wo = -(-H // s), -(-W // s)
    dy = rng.uniform(-1, 1, size=(B, Ho, Wo, Co))
    return x, w, dy


@pytest.mark.parametrize("B,H,W,Ci,Co,s", CASES)
def test_conv_fwd(B, H, W, Ci, Co, s):
    from blurred_gan_amd import ops
    x, w, dy = _data(B, H, W, Ci, Co, s)
    ref = O.conv2d_fwd(x, w, s)
    wT = dev(np.transpose(w, (0, 1, 3, 2)))
    y = ops.conv2d_fwd(dev(x), wT, torch.empty(ref.shape, device="cuda"), 5, s)
    torch.cuda.synchronize()
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-4, atol=conv_tol(25 * Ci, np.abs(ref).max()))
    # the transposed weight copy made by the library's own transpose kernel
    wT2 = ops.transpose_last2(dev(w), torch.empty(w.size, device="cuda"), 25, Ci, Co)
    assert torch.equal(wT2.view(-1), wT.reshape(-1))


@pytest.mark.parametrize("B,H,W,Ci,Co,s", CASES)
def test_conv_bwd_data(B, H, W, Ci, Co, s):
    from blurred_gan_amd import ops
    x, w, dy = _data(B, H, W, Ci, Co, s, seed=1)
    ref = O.conv2d_bwd_data(dy, w, s, (H, W))
    dx = ops.conv2d_bwd_data(dev(dy), dev(w), torch.empty(x.shape, device="cuda"), 5, s)
    torch.cuda.synchronize()
    np.testing.assert_allclose(dx.cpu().numpy(), ref, rtol=1e-4, atol=conv_tol(25 * Co, np.abs(ref).max()))


@pytest.mark.parametrize("B,H,W,Ci,Co,s", CASES)
def test_conv_bwd_filter(B, H, W, Ci, Co, s):
    from blurred_gan_amd import ops
    x, w, dy = _data(B, H, W, Ci, Co, s, seed=2)
    ref = O.conv2d_bwd_filter(x, dy, s, 5)
    nb = ops.conv2d_bwd_filter_workspace_bytes(B, H, W, Ci, Co, 5, s)
    ws = torch.empty(nb // 4 + 4, device="cuda") if nb else None
    dw = torch.full(w.shape, 7.0, device="cuda")              # poisoned: beta = 0 must overwrite
    ops.conv2d_bwd_filter(dev(x), dev(dy), dw, 5, s, 0.0, 1.0, ws)
    torch.cuda.synchronize()
    K = dy.shape[0] * dy.shape[1] * dy.shape[2]
    np.testing.assert_allclose(dw.cpu().numpy(), ref, rtol=1e-4, atol=conv_tol(K, np.abs(ref).max()))
    # accumulate form: dw = 0.5*dw + 2*grad
    ops.conv2d_bwd_filter(dev(x), dev(dy), dw, 5, s, 0.5, 2.0, ws)
    torch.cuda.synchronize()
    np.testing.assert_allclose(dw.cpu().numpy(), 2.5 * ref, rtol=1e-4, atol=2.5 * conv_tol(K, np.abs(ref).max()))


def test_transpose_batched():
    """Every conv kernel of a network transposed by one launch (float4 path and the scalar path for thin shapes)."""
    from blurred_gan_amd import ops
    shapes = [(25, 32, 64), (25, 3, 32), (25, 32, 3), (9, 100, 36), (25, 128, 256), (1, 4, 4)]
    rng = np.random.default_rng(9)
    offs, src_parts, off = [], [], 0
    for (T, R, Cc) in shapes:
        offs.append(off)
        n = T * R * Cc
        src_parts.append(rng.uniform(-1, 1, size=n))
        off += -(-n // 4) * 4
        src_parts.append(np.zeros(off - offs[-1] - n))
    src = dev(np.concatenate(src_parts))
    dst = torch.full((off,), 7.0, device="cuda")
    desc, tile = [], 0
    for o, (T, R, Cc) in zip(offs, shapes):
        desc.append([o, o, T, R, Cc, tile])
        tile += T * (-(-R // 64)) * (-(-Cc // 64))
    ops.transpose_last2_batched(src, dst, torch.tensor(desc, dtype=torch.int32, device="cuda"), len(shapes), tile)
    torch.cuda.synchronize()
    for o, (T, R, Cc) in zip(offs, shapes):
        want = src[o:o + T * R * Cc].view(T, R, Cc).permute(0, 2, 1).contiguous().view(-1)
        assert torch.equal(dst[o:o + T * R * Cc], want), (T, R, Cc)


def test_epilogues():
    from blurred_gan_amd import ops
    from blurred_gan_amd._lib import EPI_BIAS_LRELU, EPI_MUL_GRAD, EPI_TANH, EPI_NONE
    B, H, W, Ci, Co, s = 2, 8, 8, 32, 32, 2
    x, w, dy = _data(B, H, W, Ci, Co, s, seed=3)
    rng = np.random.default_rng(4)
    bias = rng.normal(size=Co)
    keep = (rng.uniform(size=dy.shape) >= 0.3).astype(np.uint8)
    z = O.conv2d_fwd(x, w, s) + bias
    wT = dev(np.transpose(w, (0, 1, 3, 2)))
    out = lambda: torch.empty(z.shape, device="cuda")
    tol = conv_tol(25 * Ci, np.abs(z).max())
    y = ops.conv2d_fwd(dev(x), wT, out(), 5, s, ops.epilogue(EPI_BIAS_LRELU, bias=dev(bias), keep=dev(keep, torch.uint8), alpha=0.3, scale=1 / 0.7))
    np.testing.assert_allclose(y.cpu().numpy(), O.dropout_fwd(O.lrelu_fwd(z), keep, 0.3), rtol=1e-4, atol=2 * tol)
    y = ops.conv2d_fwd(dev(x), wT, out(), 5, s, ops.epilogue(EPI_BIAS_LRELU, bias=dev(bias), alpha=0.3))
    np.testing.assert_allclose(y.cpu().numpy(), O.lrelu_fwd(z), rtol=1e-4, atol=tol)
    y = ops.conv2d_fwd(dev(x), wT, out(), 5, s, ops.epilogue(EPI_TANH, bias=dev(bias)))
    np.testing.assert_allclose(y.cpu().numpy(), np.tanh(z), rtol=1e-4, atol=tol)
    y = ops.conv2d_fwd(dev(x), wT, out(), 5, s, ops.epilogue(EPI_NONE, bias=dev(bias)))
    np.testing.assert_allclose(y.cpu().numpy(), z, rtol=1e-4, atol=tol)
    ref_act = rng.normal(size=x.shape)
    keep_x = (rng.uniform(size=x.shape) >= 0.3).astype(np.uint8)
    dxr = O.conv2d_bwd_data(dy, w, s, (H, W))
    dx = ops.conv2d_bwd_data(dev(dy), dev(w), torch.empty(x.shape, device="cuda"), 5, s,
                             ops.epilogue(EPI_MUL_GRAD, ref=dev(ref_act), keep=dev(keep_x, torch.uint8), alpha=0.3, scale=1 / 0.7))
    exp = dxr * O.lrelu_mask(ref_act) * keep_x / 0.7
    np.testing.assert_allclose(dx.cpu().numpy(), exp, rtol=1e-4, atol=2 * conv_tol(25 * Co, np.abs(dxr).max()))


@pytest.mark.parametrize("B,H,W,Ci,Co,s", [
    (4, 16, 16, 32, 64, 2),       # gather-GEMM, float4 epilogue
    (130, 4, 4, 128, 256, 2),     # position-major tiles + split-K (slab reduce writes the output)
    (3, 64, 64, 16, 32, 2),       # row-staged 16-channel forward
    (2, 16, 16, 3, 32, 2),        # thin-K row kernel
    (2, 9, 7, 20, 12, 2),         # unaligned: direct kernel
    (2, 16, 16, 32, 3, 1),        # thin-N row kernel
])
def test_mul_grad_epilogue_with_ref_aliasing_the_output(B, H, W, Ci, Co, s):
    """include/bgan.h BG_EPI_MUL_GRAD: `ref` may alias the output -- the penalty's linearised forward (engine.Net.gp_second_order_merged)
    overwrites the activation rows whose signs it consumes.  Every forward kernel family must read ref[i] in the thread that stores
    y[i], before that store: in place == out of place, bit for bit."""
    from blurred_gan_amd import ops
    from blurred_gan_amd._lib import EPI_MUL_GRAD
    x, w, dy = _data(B, H, W, Ci, Co, s, seed=21)
    wT = dev(np.transpose(w, (0, 1, 3, 2)))
    Ho, Wo = -(-H // s), -(-W // s)
    act = torch.randn(B, Ho, Wo, Co, device="cuda")
    nb = ops.conv2d_splitk_workspace_bytes(False, B, H, W, Ci, Co, 5, s)
    ws = torch.empty(nb // 4 + 4, device="cuda") if nb else None
    sep = ops.conv2d_fwd(dev(x), wT, torch.empty_like(act), 5, s, ops.epilogue(EPI_MUL_GRAD, ref=act.clone(), alpha=0.3, ws=ws))
    inplace = act.clone()
    ops.conv2d_fwd(dev(x), wT, inplace, 5, s, ops.epilogue(EPI_MUL_GRAD, ref=inplace, alpha=0.3, ws=ws))
    assert torch.equal(sep, inplace)
    ref = O.conv2d_fwd(x, w, s) * np.where(act.cpu().numpy() > 0, 1.0, 0.3)
    np.testing.assert_allclose(sep.cpu().numpy(), ref, rtol=1e-4, atol=2 * conv_tol(25 * Ci, np.abs(ref).max()))


def test_c16_epilogues():
    """The row-staged 16-channel data gradient with the critic's fused LeakyReLU'/dropout product, mask on the leading samples only."""
    from blurred_gan_amd import ops
    from blurred_gan_amd._lib import EPI_MUL_GRAD, EPI_BIAS_LRELU
    B, H, W, Ci, Co, s = 3, 64, 64, 16, 32, 2
    x, w, dy = _data(B, H, W, Ci, Co, s, seed=7)
    rng = np.random.default_rng(8)
    ref_act = rng.normal(size=x.shape)
    keep_x = (rng.uniform(size=x.shape) >= 0.3).astype(np.uint8)
    dxr = O.conv2d_bwd_data(dy, w, s, (H, W))
    n_keep = 2 * H * W * Ci
    dx = ops.conv2d_bwd_data(dev(dy), dev(w), torch.empty(x.shape, device="cuda"), 5, s,
                             ops.epilogue(EPI_MUL_GRAD, ref=dev(ref_act), keep=dev(keep_x, torch.uint8), alpha=0.3, scale=1 / 0.7, keep_elems=n_keep))
    exp = dxr * O.lrelu_mask(ref_act)
    exp[:2] *= keep_x[:2] / 0.7
    tol = conv_tol(25 * Co, np.abs(dxr).max())
    np.testing.assert_allclose(dx.cpu().numpy(), exp, rtol=1e-4, atol=2 * tol)
    bias = rng.normal(size=Ci)
    y = ops.conv2d_bwd_data(dev(dy), dev(w), torch.empty(x.shape, device="cuda"), 5, s, ops.epilogue(EPI_BIAS_LRELU, bias=dev(bias), alpha=0.3))
    np.testing.assert_allclose(y.cpu().numpy(), O.lrelu_fwd(dxr + bias), rtol=1e-4, atol=2 * tol)
    # forward 16 -> 32 with the critic's fused bias + LeakyReLU + dropout (mask on the first two samples), and with tanh (generic path)
    from blurred_gan_amd._lib import EPI_TANH
    bias_o = rng.normal(size=Co)
    z = O.conv2d_fwd(x, w, s) + bias_o
    keep_y = (rng.uniform(size=z.shape) >= 0.3).astype(np.uint8)
    wT = dev(np.transpose(w, (0, 1, 3, 2)))
    tolf = conv_tol(25 * Ci, np.abs(z).max())
    yk = ops.conv2d_fwd(dev(x), wT, torch.empty(z.shape, device="cuda"), 5, s,
                        ops.epilogue(EPI_BIAS_LRELU, bias=dev(bias_o), keep=dev(keep_y, torch.uint8), alpha=0.3, scale=1 / 0.7, keep_elems=2 * z[0].size))
    expf = O.lrelu_fwd(z)
    expf[:2] = expf[:2] * keep_y[:2] / 0.7
    np.testing.assert_allclose(yk.cpu().numpy(), expf, rtol=1e-4, atol=2 * tolf)
    yt = ops.conv2d_fwd(dev(x), wT, torch.empty(z.shape, device="cuda"), 5, s, ops.epilogue(EPI_TANH, bias=dev(bias_o)))
    np.testing.assert_allclose(yt.cpu().numpy(), np.tanh(z), rtol=1e-4, atol=tolf)


@pytest.mark.parametrize("B,HW", [(5, 16), (3, 32), (2, 64)])
def test_thin_n_row_kernel_epilogues(B, HW):
    """conv_rows_scatter_kernel (<= 3 channels out): its three epilogue variants -- bias only, bias + tanh (branch-free tanh), and
    the generic one with per-element loads -- on workgroups that hold 4 / 2 / 1 images (B = 5 at 16 pixels leaves a group with one
    image: the rest of its buffer is out of range), forward and data gradient, masks on the leading samples only."""
    from blurred_gan_amd import ops
    from blurred_gan_amd._lib import EPI_BIAS_LRELU, EPI_MUL_GRAD, EPI_TANH, EPI_NONE, EPI_AFFINE_LRELU
    rng = np.random.default_rng(100 + HW)
    # forward 32 -> 3, stride 1
    Ci, Co, s = 32, 3, 1
    x, w, _ = _data(B, HW, HW, Ci, Co, s, seed=HW)
    bias = rng.normal(size=Co)
    z = O.conv2d_fwd(x, w, s) + bias
    wT = dev(np.transpose(w, (0, 1, 3, 2)))
    out = lambda: torch.empty(z.shape, device="cuda")
    tol = conv_tol(25 * Ci, np.abs(z).max())
    y = ops.conv2d_fwd(dev(x), wT, out(), 5, s, ops.epilogue(EPI_NONE, bias=dev(bias)))
    np.testing.assert_allclose(y.cpu().numpy(), z, rtol=1e-4, atol=tol)
    y = ops.conv2d_fwd(dev(x), wT, out(), 5, s, ops.epilogue(EPI_TANH, bias=dev(bias)))
    np.testing.assert_allclose(y.cpu().numpy(), np.tanh(z), rtol=1e-4, atol=tol)
    keep = (rng.uniform(size=z.shape) >= 0.3).astype(np.uint8)
    nk = (B - 1) * z[0].size                                    # the mask covers all samples but the last
    y = ops.conv2d_fwd(dev(x), wT, out(), 5, s, ops.epilogue(EPI_BIAS_LRELU, bias=dev(bias), keep=dev(keep, torch.uint8), alpha=0.3,
                                                              scale=1 / 0.7, keep_elems=nk))
    exp = O.lrelu_fwd(z)
    exp[:B - 1] = exp[:B - 1] * keep[:B - 1] / 0.7
    np.testing.assert_allclose(y.cpu().numpy(), exp, rtol=1e-4, atol=2 * tol)
    mul = rng.uniform(0.5, 1.5, size=Co)
    z0 = O.conv2d_fwd(x, w, s)
    y = ops.conv2d_fwd(dev(x), wT, out(), 5, s, ops.epilogue(EPI_AFFINE_LRELU, bias=dev(bias), ref=dev(mul), alpha=0.3))
    np.testing.assert_allclose(y.cpu().numpy(), O.lrelu_fwd(z0 * mul + bias), rtol=1e-4, atol=2 * tol)
    # data gradient of a stride-2 conv 3 -> 32 (dy has 32 channels, dx 3): LeakyReLU' x mask of the layer input, and plain
    Ci, Co, s = 3, 32, 2
    x, w, dy = _data(B, 2 * HW, 2 * HW, Ci, Co, s, seed=HW + 1)
    dxr = O.conv2d_bwd_data(dy, w, s, (2 * HW, 2 * HW))
    told = conv_tol(25 * Co, np.abs(dxr).max())
    dx = ops.conv2d_bwd_data(dev(dy), dev(w), torch.empty(x.shape, device="cuda"), 5, s, None)
    np.testing.assert_allclose(dx.cpu().numpy(), dxr, rtol=1e-4, atol=told)
    ref_act = rng.normal(size=x.shape)
    keep_x = (rng.uniform(size=x.shape) >= 0.3).astype(np.uint8)
    dx = ops.conv2d_bwd_data(dev(dy), dev(w), torch.empty(x.shape, device="cuda"), 5, s,
                             ops.epilogue(EPI_MUL_GRAD, ref=dev(ref_act), keep=dev(keep_x, torch.uint8), alpha=0.3, scale=1 / 0.7,
                                          keep_elems=(B - 1) * x[0].size))
    expd = dxr * O.lrelu_mask(ref_act)
    expd[:B - 1] *= keep_x[:B - 1] / 0.7
    np.testing.assert_allclose(dx.cpu().numpy(), expd, rtol=1e-4, atol=2 * told)


def test_unaligned_epilogue_operands_take_the_direct_kernel():
    """x / w / y must be 16-byte aligned (the call fails loudly otherwise); the epilogue's operands need not be.  The matrix-core
    kernels read them four at a time (include/bgan.h), so a reference tensor that starts 4 bytes into an allocation or a mask that
    starts at an odd byte sends the call to the direct kernel -- same results."""
    from blurred_gan_amd import ops
    from blurred_gan_amd._lib import EPI_BIAS_LRELU, EPI_MUL_GRAD
    B, H, W, Ci, Co, s = 3, 16, 16, 32, 64, 2
    x, w, dy = _data(B, H, W, Ci, Co, s, seed=21)
    rng = np.random.default_rng(22)
    bias = rng.normal(size=Co)
    z = O.conv2d_fwd(x, w, s) + bias
    keep = (rng.uniform(size=z.shape) >= 0.3).astype(np.uint8)
    wT = dev(np.transpose(w, (0, 1, 3, 2)))
    exp = O.dropout_fwd(O.lrelu_fwd(z), keep, 0.3)
    tol = conv_tol(25 * Ci, np.abs(z).max())
    ref_act = rng.normal(size=x.shape)
    keep_x = (rng.uniform(size=x.shape) >= 0.3).astype(np.uint8)
    dxr = O.conv2d_bwd_data(dy, w, s, (H, W))
    expd = dxr * O.lrelu_mask(ref_act) * keep_x / 0.7
    told = conv_tol(25 * Co, np.abs(dxr).max())

    def shifted(a, dtype, shift):
        buf = torch.empty(a.size + 8, device="cuda", dtype=dtype)
        v = buf[shift:shift + a.size].view(a.shape)
        v.copy_(dev(a, dtype))
        return v

    for shift in (0, 1):
        y = ops.conv2d_fwd(dev(x), wT, torch.empty(z.shape, device="cuda"), 5, s,
                           ops.epilogue(EPI_BIAS_LRELU, bias=shifted(bias, torch.float32, shift), keep=shifted(keep, torch.uint8, shift),
                                        alpha=0.3, scale=1 / 0.7))
        np.testing.assert_allclose(y.cpu().numpy(), exp, rtol=1e-4, atol=2 * tol)
        dx = ops.conv2d_bwd_data(dev(dy), dev(w), torch.empty(x.shape, device="cuda"), 5, s,
                                 ops.epilogue(EPI_MUL_GRAD, ref=shifted(ref_act, torch.float32, shift), keep=shifted(keep_x, torch.uint8, shift),
                                              alpha=0.3, scale=1 / 0.7))
        np.testing.assert_allclose(dx.cpu().numpy(), expd, rtol=1e-4, atol=2 * told)
    ybuf = torch.empty(z.size + 4, device="cuda")
    with pytest.raises(ValueError, match="16-byte aligned"):
        ops.conv2d_fwd(dev(x), wT, ybuf[1:1 + z.size].view(z.shape), 5, s)


def test_thin_n_row_kernel_tanh_keeps_nan_and_saturates():
    """The branch-free tanh of the thin-N row kernel: a NaN in the pre-activation stays a NaN (as with tanhf), huge values give +-1."""
    from blurred_gan_amd import ops
    from blurred_gan_amd._lib import EPI_TANH
    B, HW, Ci, Co = 1, 16, 32, 3
    x = np.zeros((B, HW, HW, Ci))
    w = np.zeros((5, 5, Ci, Co))
    w[2, 2, 0, :] = 1.0                                           # y[..., n] = x[..., 0]
    x[0, 2, 2, 0] = np.nan                                        # (0 * NaN = NaN reaches the 5 x 5 neighbourhood: keep the probes apart)
    x[0, 8, 8, 0] = 1e30
    x[0, 12, 12, 0] = -1e30
    x[0, 8, 13, 0] = 0.05
    wT = dev(np.transpose(w, (0, 1, 3, 2)))
    y = ops.conv2d_fwd(dev(x), wT, torch.empty((B, HW, HW, Co), device="cuda"), 5, 1, ops.epilogue(EPI_TANH)).cpu().numpy()
    assert np.isnan(y[0, 2, 2]).all()
    assert (y[0, 8, 8] == 1.0).all() and (y[0, 12, 12] == -1.0).all()
    np.testing.assert_allclose(y[0, 8, 13], np.tanh(0.05), rtol=2e-6)
    assert np.isfinite(y[0, 6:]).all()


def test_conv_transpose_roles():
    """Conv2DTranspose forward = bwd_data with the kernel array as is; its filter gradient swaps x and dy."""
    from blurred_gan_amd import ops
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, size=(2, 4, 4, 32))
    w = rng.uniform(-1, 1, size=(5, 5, 16, 32)) / 20           # [k,k,c_out,c_in]
    ref = O.conv2d_transpose_fwd(x, w, 2)
    y = ops.conv2d_bwd_data(dev(x), dev(w), torch.empty(ref.shape, device="cuda"), 5, 2)
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-4, atol=conv_tol(800, np.abs(ref).max()))
    dyt = rng.uniform(-1, 1, size=ref.shape)
    dwr = O.conv2d_transpose_bwd_filter(x, dyt, 2, 5)
    nb = ops.conv2d_bwd_filter_workspace_bytes(2, 8, 8, 16, 32, 5, 2)
    ws = torch.empty(nb // 4 + 4, device="cuda") if nb else None
    dw = ops.conv2d_bwd_filter(dev(dyt), dev(x), torch.empty(w.shape, device="cuda"), 5, 2, 0.0, 1.0, ws)
    np.testing.assert_allclose(dw.cpu().numpy(), dwr, rtol=1e-4, atol=conv_tol(32, np.abs(dwr).max()))


@pytest.mark.parametrize("B,H,W,Ci,Co,s", [(256, 32, 32, 32, 64, 2), (256, 64, 64, 3, 32, 2), (256, 4, 4, 256, 512, 2),
                                           (256, 64, 64, 32, 3, 1), (256, 16, 16, 128, 64, 2),
                                           (128, 128, 128, 16, 32, 2), (384, 64, 64, 16, 32, 2)])
def test_adjointness_full_size(B, H, W, Ci, Co, s):
    """<conv(x), dy> == <x, conv^T(dy)> == <w, wgrad(x, dy)> at the C2 layer sizes (no oracle needed)."""
    from blurred_gan_amd import ops
    torch.manual_seed(0)
    x = torch.rand(B, H, W, Ci, device="cuda") * 2 - 1
    w = (torch.rand(5, 5, Ci, Co, device="cuda") * 2 - 1) / (25 * Ci) ** 0.5
    Ho, Wo = -(-H // s), -(-W // s)
    dy = torch.rand(B, Ho, Wo, Co, device="cuda") * 2 - 1
    wT = ops.transpose_last2(w, torch.empty_like(w).view(-1), 25, Ci, Co)
    y = ops.conv2d_fwd(x, wT, torch.empty(B, Ho, Wo, Co, device="cuda"), 5, s)
    dx = ops.conv2d_bwd_data(dy, w, torch.empty_like(x), 5, s)
    nb = ops.conv2d_bwd_filter_workspace_bytes(B, H, W, Ci, Co, 5, s)
    ws = torch.empty(nb // 4 + 4, device="cuda") if nb else None
    dw = ops.conv2d_bwd_filter(x, dy, torch.empty_like(w), 5, s, 0.0, 1.0, ws)
    a = (y.double() * dy.double()).sum().item()
    b = (x.double() * dx.double()).sum().item()
    c = (w.double() * dw.double()).sum().item()
    scale = max(1.0, abs(a))
    assert abs(a - b) < 2e-4 * scale and abs(a - c) < 2e-4 * scale, (a, b, c)


def test_conv_rejects_bad_arguments():
    from blurred_gan_amd import ops
    x = torch.zeros(1, 4, 4, 32, device="cuda")
    w = torch.zeros(9 * 32 * 32, device="cuda")
    with pytest.raises(ValueError):
        ops.conv2d_fwd(x, torch.zeros(49 * 32 * 32, device="cuda"), torch.zeros(1, 4, 4, 32, device="cuda"), 7, 1)   # k*k > 25
    with pytest.raises(ValueError):
        ops.conv2d_fwd(x, w, torch.zeros(1, 2, 2, 32, device="cuda"), 3, 3)                                          # stride 3


def _random_cases(n, seed):
    rng = np.random.default_rng(seed)
    chans = [1, 2, 3, 4, 8, 12, 16, 20, 32, 48, 64, 96, 128]
    out = []
    while len(out) < n:
        B = int(rng.integers(1, 6))
        H, W = int(rng.integers(2, 41)), int(rng.integers(2, 41))
        if rng.uniform() < 0.3:                                   # the row kernels want widths that are multiples of 16
            W = int(rng.choice([16, 32, 64]))
        Ci, Co = int(rng.choice(chans)), int(rng.choice(chans))
        s = int(rng.choice([1, 2]))
        if B * H * W * max(Ci, Co) > 400_000:
            continue
        out.append((B, H, W, Ci, Co, s))
    return out


@pytest.mark.parametrize("B,H,W,Ci,Co,s", _random_cases(48, 20261004))
def test_conv_random_shapes(B, H, W, Ci, Co, s):
    """Seeded random geometries across the dispatch boundaries (thin / 16-channel / MFMA / direct paths, odd sizes, both strides):
    forward, data gradient and filter gradient against the float64 oracle."""
    from blurred_gan_amd import ops
    x, w, dy = _data(B, H, W, Ci, Co, s, seed=B * 1000 + H * 31 + W)
    wT = dev(np.transpose(w, (0, 1, 3, 2)))
    ref = O.conv2d_fwd(x, w, s)
    y = ops.conv2d_fwd(dev(x), wT, torch.empty(ref.shape, device="cuda"), 5, s)
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-4, atol=conv_tol(25 * Ci, np.abs(ref).max()))
    refd = O.conv2d_bwd_data(dy, w, s, (H, W))
    dx = ops.conv2d_bwd_data(dev(dy), dev(w), torch.empty(x.shape, device="cuda"), 5, s)
    np.testing.assert_allclose(dx.cpu().numpy(), refd, rtol=1e-4, atol=conv_tol(25 * Co, np.abs(refd).max()))
    refw = O.conv2d_bwd_filter(x, dy, s, 5)
    nb = ops.conv2d_bwd_filter_workspace_bytes(B, H, W, Ci, Co, 5, s)
    ws = torch.empty(nb // 4 + 4, device="cuda") if nb else None
    dw = ops.conv2d_bwd_filter(dev(x), dev(dy), torch.empty(w.shape, device="cuda"), 5, s, 0.0, 1.0, ws)
    np.testing.assert_allclose(dw.cpu().numpy(), refw, rtol=1e-4, atol=conv_tol(dy[..., 0].size, np.abs(refw).max()))


@pytest.mark.parametrize("B,H,W,Ci,Co,s,bwd", [
    (8, 32, 32, 64, 128, 2, True), (8, 64, 64, 32, 64, 2, True),       # ConvT 128->64 and 64->32 as data gradients: 64x64 and 128x32 tiles, 4 phases
    (130, 8, 8, 128, 256, 2, True),                                       # position-major tiles, padding taps skipped, ragged last M tile
    (1024, 16, 16, 16, 32, 2, True),                                      # four phases merged in one workgroup
    (3, 16, 16, 64, 128, 1, False), (5, 9, 7, 32, 64, 2, False),          # forward, stride 1 / odd sizes (phases of different extents)
])
def test_conv_epilogue_leaves_batchnorm_statistics(B, H, W, Ci, Co, s, bwd):
    """bg_epilogue.stats: the gather-GEMM also writes one row of column sums / sums of squares per workgroup; summed over the rows
    they equal the statistics of the tensor it stored (what the BatchNormalization behind a Conv2DTranspose needs,
    demo_celeba.py:62-90).  A geometry that takes another kernel family reports 0 rows."""
    from blurred_gan_amd import ops
    x, w, dy = _data(B, H, W, Ci, Co, s, seed=3)
    if bwd:
        src, wd = dev(dy), dev(w).reshape(25, Ci, Co)
        out = torch.empty(B, H, W, Ci, device="cuda")
        N = Ci
    else:
        wT = ops.transpose_last2(dev(w), torch.empty(w.size, device="cuda"), 25, Ci, Co)
        src = dev(x)
        out = torch.empty(B, -(-H // s), -(-W // s), Co, device="cuda")
        N = Co
    stats = torch.full(((out.numel() // N // 32 + 64) * 2 * N,), float("nan"), device="cuda")
    epi = ops.epilogue(stats=stats)
    if bwd:
        ops.conv2d_bwd_data(src, wd, out, 5, s, epi)
    else:
        ops.conv2d_fwd(src, wT, out, 5, s, epi)
    rows = ops.conv2d_stats_rows(epi)
    assert rows > 0, "this geometry is expected on the MFMA gather kernel without split-K"
    part = stats[:rows * 2 * N].view(rows, 2, N).double().cpu().numpy()
    assert np.isfinite(part).all(), "every partial row must have been written"
    flat = out.view(-1, N).double().cpu().numpy()
    scale = np.abs(flat).max()
    np.testing.assert_allclose(part[:, 0].sum(0), flat.sum(0), rtol=1e-5, atol=2e-5 * scale * np.sqrt(flat.shape[0]))
    np.testing.assert_allclose(part[:, 1].sum(0), (flat ** 2).sum(0), rtol=1e-5, atol=1e-6)
    # a thin layer goes to another kernel family: no rows, the caller runs the normal statistics pass
    o2 = torch.empty(2, 16, 16, 3, device="cuda")
    epi2 = ops.epilogue(stats=stats)
    ops.conv2d_fwd(torch.rand(2, 16, 16, 32, device="cuda"), torch.rand(25 * 3 * 32, device="cuda"), o2, 5, 1, epi2)
    assert ops.conv2d_stats_rows(epi2) == 0 and ops.conv2d_stats_rows(epi) == rows      # per call, not last-call state
