"""GPU parity: bg_blur_nhwc_f32 (through the C ABI) vs the float64 oracle."""
import numpy as np
import pytest
import torch

from oracle import np_ops as O
from helpers import dev, POINT_RTOL, POINT_ATOL

pytestmark = pytest.mark.gpu


def _run(x, std):
    from blurred_gan_amd import ops
    B, H, W, C = x.shape
    ks, se, nt = ops.blur_policy(std, H, W)
    taps = torch.tensor(ops.gauss_kernel_1d(se, ks), device="cuda")
    nb = ops.blur_workspace_bytes(B, H, W, C, nt)
    tmp = torch.empty(nb // 4 + 4, device="cuda") if nb else None
    y = ops.blur_nhwc(dev(x), torch.empty(x.shape, device="cuda"), taps, nt, tmp)
    torch.cuda.synchronize()
    return y.cpu().numpy(), (ks, se, nt)


@pytest.mark.parametrize("shape,std", [
    ((3, 8, 8, 3), 0.05), ((2, 28, 28, 1), 0.05), ((2, 28, 28, 1), 23.5), ((4, 64, 64, 3), 5.0), ((2, 64, 64, 3), 4.94),
    ((2, 64, 64, 3), 0.5), ((2, 9, 13, 3), 1.0), ((1, 128, 128, 3), 5.0), ((2, 128, 128, 3), 23.5), ((1, 256, 256, 3), 23.5),
    ((1, 256, 256, 3), 42.34), ((2, 12, 12, 1), 0.7), ((2, 5, 7, 2), 2.0),
    ((2, 100, 72, 3), 3.0), ((1, 218, 178, 3), 5.0), ((3, 96, 80, 1), 8.0), ((2, 130, 66, 2), 30.0), ((1, 218, 178, 3), 40.0),
    ((2, 128, 128, 4), 5.0), ((3, 160, 144, 1), 6.0), ((2, 128, 128, 3), 1.0), ((2, 192, 136, 2), 2.0),   # band passes at narrow kernels, 1 / 2 / 4 channels   # line kernels / banded Toeplitz passes (>= 100 taps), ragged blocks
    # fused streaming strips (1 / 3 channels, <= 65 taps, larger than 64 pixels): 3..65 taps, both halo widths, heights that are
    # not a multiple of 16 / 32, widths that end in a partial strip, more images than XCDs and fewer
    ((2, 256, 256, 3), 5.0), ((1, 256, 256, 3), 10.5), ((1, 256, 256, 3), 5.4), ((11, 72, 40, 1), 5.0), ((3, 300, 260, 3), 4.0),
    ((2, 66, 100, 3), 0.05), ((9, 130, 68, 3), 2.0), ((2, 257, 96, 1), 10.0), ((1, 90, 28, 1), 1.0),
    # band passes above 65 taps with wave-private tiles: a ragged last column group, tiles that start / end inside a pixel
    # (3 channels), row groups that end in a partial 32-row block, 1 / 3 / 4 channels
    ((2, 200, 140, 3), 23.5), ((1, 136, 150, 4), 20.0), ((2, 150, 260, 1), 30.0), ((1, 72, 332, 3), 12.0), ((1, 330, 68, 3), 42.0),
    # widths / heights whose rows are not float4-addressable in one pass, the other, or both (dword chunk loads, scalar stores)
    ((2, 256, 250, 3), 23.5), ((2, 250, 256, 3), 23.5), ((1, 250, 250, 3), 23.5), ((2, 150, 131, 1), 30.0), ((1, 133, 140, 2), 20.0),
])
def test_blur_matches_oracle(shape, std):
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, size=shape).astype(np.float32)
    y, (ks, se, nt) = _run(x, std)
    oks, ose, ont = O.blur_policy(std, shape[1], shape[2])
    assert (ks, nt) == (oks, ont) and abs(se - ose) < 1e-6 * max(1, ose)     # host policy == oracle policy
    ref = O.blur_images(x.astype(np.float64), std)
    np.testing.assert_allclose(y, ref, rtol=POINT_RTOL, atol=POINT_ATOL * 4)


@pytest.mark.parametrize("ld", ["0", "1"])
@pytest.mark.parametrize("shape,std", [((2, 200, 140, 3), 23.5), ((1, 136, 152, 4), 20.0), ((2, 150, 260, 1), 30.0)])
def test_band_passes_slower_loaders(shape, std, ld, monkeypatch):
    """The band passes' fallback chunk loaders -- dword buffer loads (1) and the synchronous 64-bit path kept for images beyond
    32-bit byte offsets (0) -- forced on shapes that would take the float4 loader."""
    monkeypatch.setenv("BG_BLUR_BAND_LD", ld)
    x = np.random.default_rng(3).uniform(-1, 1, size=shape).astype(np.float32)
    y, _ = _run(x, std)
    np.testing.assert_allclose(y, O.blur_images(x.astype(np.float64), std), rtol=POINT_RTOL, atol=POINT_ATOL * 4)


@pytest.mark.parametrize("shape,std", [((5, 64, 64, 3), 5.0), ((3, 64, 64, 3), 10.5), ((2, 64, 64, 1), 5.0), ((3, 28, 28, 1), 9.0),
                                       ((2, 48, 40, 3), 3.0), ((3, 33, 64, 3), 2.0), ((2, 64, 36, 1), 4.0), ((2, 40, 24, 2), 3.0),
                                       ((2, 60, 32, 4), 2.5), ((1, 17, 4, 1), 2.0)])
def test_small_image_kernels_row_blocks_and_whole_image(shape, std, monkeypatch):
    """Images up to 64 x 64 at >= 13 taps: the row-block kernel (blur_rows_kernel: 32 output rows per workgroup, NHWC-interleaved
    Toeplitz products; float4-addressable rows, <= 4 channels) and, forced, the whole-image kernel it replaced -- one and two
    row blocks, partial last block, rows narrower than a column tile, every channel count."""
    x = np.random.default_rng(11).uniform(-1, 1, size=shape).astype(np.float32)
    ref = O.blur_images(x.astype(np.float64), std)
    assert O.blur_policy(std, shape[1], shape[2])[2] >= 13
    y, _ = _run(x, std)
    np.testing.assert_allclose(y, ref, rtol=POINT_RTOL, atol=POINT_ATOL * 4)
    monkeypatch.setenv("BG_BLUR_NO_ROWS", "1")
    y0, _ = _run(x, std)
    np.testing.assert_allclose(y0, ref, rtol=POINT_RTOL, atol=POINT_ATOL * 4)


@pytest.mark.parametrize("shape,std", [((3, 96, 224, 3), 15.0), ((5, 256, 128, 3), 30.0), ((2, 160, 256, 3), 12.0), ((9, 128, 96, 3), 23.5),
                                       ((2, 256, 256, 3), 23.5), ((2, 256, 256, 3), 42.34), ((1, 512, 64, 3), 20.0), ((4, 128, 128, 3), 11.2),
                                       ((8, 96, 128, 3), 12.0), ((16, 160, 96, 3), 14.0), ((40, 96, 96, 3), 12.0)])
def test_wide_tap_panel_kernel(shape, std, monkeypatch):
    """blur_panel16_kernel / blur_panel_kernel (> 65 taps, RGB, 16-row panels up to 208 taps and 32-row panels beyond or with
    BG_BLUR_PANEL16=0, both passes in one launch, the pass-1 result in LDS): odd and even panel counts, non-square images, panels whose band is clipped on one side / both sides / not at all, bands wider than the image,
    batches that are multiples of 8 (the XCD-aware placement, with and without its reversed image groups) --
    against the float64 oracle, and against the two-launch band passes it replaces (BG_BLUR_NO_PANEL=1)."""
    x = np.random.default_rng(5).uniform(-1, 1, size=shape).astype(np.float32)
    assert O.blur_policy(std, shape[1], shape[2])[2] >= 67
    ref = O.blur_images(x.astype(np.float64), std)
    y, _ = _run(x, std)
    np.testing.assert_allclose(y, ref, rtol=POINT_RTOL, atol=POINT_ATOL * 4)
    monkeypatch.setenv("BG_BLUR_PANEL16", "0")                   # the 32-row kernel on every case
    y32, _ = _run(x, std)
    np.testing.assert_allclose(y32, ref, rtol=POINT_RTOL, atol=POINT_ATOL * 4)
    np.testing.assert_allclose(y, y32, rtol=2e-5, atol=2e-6)
    monkeypatch.setenv("BG_BLUR_NO_PANEL", "1")
    y0, _ = _run(x, std)
    np.testing.assert_allclose(y0, ref, rtol=POINT_RTOL, atol=POINT_ATOL * 4)
    np.testing.assert_allclose(y, y0, rtol=2e-5, atol=2e-6)


def _random_blur_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        B, H, W = int(rng.integers(1, 4)), int(rng.integers(3, 200)), int(rng.integers(3, 200))
        C = int(rng.choice([1, 2, 3, 4, 5, 8]))
        if B * H * W * C > 300_000:
            continue
        out.append(((B, H, W, C), float(rng.choice([0.05, 0.4, 0.9, 1.7, 3.0, 5.0, 9.0, 15.0, 23.5, 40.0]))))
    return out


@pytest.mark.parametrize("shape,std", _random_blur_cases(32, 4242))
def test_blur_random_shapes(shape, std):
    """Seeded random image sizes, channel counts and sigmas across the kernel-family boundaries (whole-image MFMA / sliding
    window, band passes, line kernels for more than 4 channels)."""
    x = np.random.default_rng(7).uniform(-1, 1, size=shape).astype(np.float32)
    y, _ = _run(x, std)
    np.testing.assert_allclose(y, O.blur_images(x.astype(np.float64), std), rtol=POINT_RTOL, atol=POINT_ATOL * 4)


def test_gauss_kernel_host_matches_oracle():
    from blurred_gan_amd import ops
    for std, hw in [(0.05, 28), (1.0, 64), (5.0, 64), (4.94, 64), (23.5, 256), (42.34, 256)]:
        ks, se, nt = ops.blur_policy(std, hw, hw)
        g = np.array(ops.gauss_kernel_1d(se, ks), np.float32)
        ref = O.gaussian_kernel_1d(se, ks, np.float64)
        assert g.shape == ref.shape
        np.testing.assert_allclose(g, ref, rtol=2e-6, atol=1e-9)


def test_blur_properties_full_size():
    """Size-independent properties at BASELINE.json's C2 size: self-adjointness, linearity, constant image."""
    from blurred_gan_amd import ops
    torch.manual_seed(0)
    B, H, W, C = 256, 64, 64, 3
    ks, se, nt = ops.blur_policy(5.0, H, W)
    taps = torch.tensor(ops.gauss_kernel_1d(se, ks), device="cuda")
    x = torch.rand(B, H, W, C, device="cuda") * 2 - 1
    y = torch.rand(B, H, W, C, device="cuda") * 2 - 1
    bl = lambda t: ops.blur_nhwc(t.contiguous(), torch.empty_like(t), taps, nt)
    bx, by = bl(x), bl(y)
    lhs, rhs = (bx.double() * y.double()).sum().item(), (x.double() * by.double()).sum().item()
    assert abs(lhs - rhs) < 1e-5 * max(1.0, abs(lhs))
    z = bl(2.0 * x + 3.0 * y)
    assert (z - (2.0 * bx + 3.0 * by)).abs().max().item() < 2e-5
    ones = bl(torch.ones(2, H, W, C, device="cuda"))
    assert abs(ones[0, H // 2, W // 2, 0].item() - 1.0) < 1e-5          # interior preserved
    g = np.array(ops.gauss_kernel_1d(se, ks), np.float64)
    assert abs(ones[0, 0, 0, 1].item() - g[nt // 2:].sum() ** 2) < 1e-5  # corner darkened by the truncated sums


@pytest.mark.parametrize("shape,std", [((128, 128, 128, 3), 5.0), ((64, 256, 256, 3), 5.0), ((64, 256, 256, 3), 23.5), ((64, 256, 256, 3), 42.34)])
def test_blur_properties_c4_c5_sizes(shape, std):
    """The same size-independent properties at BASELINE.json's C4 (128x128, batch 128) and C5 (256x256, 64 per GPU; 31 / 143 / 255
    taps) sizes: the fused streaming strips at 31 taps, the two transposing band passes through the scratch image above 65."""
    from blurred_gan_amd import ops
    torch.manual_seed(1)
    B, H, W, C = shape
    ks, se, nt = ops.blur_policy(std, H, W)
    taps = torch.tensor(ops.gauss_kernel_1d(se, ks), device="cuda")
    tmp = torch.empty(shape, device="cuda")
    assert ops.blur_workspace_bytes(B, H, W, C, nt) == 0          # <= 65 taps: fused strips; above: 32-row panels -- no scratch image either way
    x = torch.rand(shape, device="cuda") * 2 - 1
    y = torch.rand(shape, device="cuda") * 2 - 1
    bl = lambda t: ops.blur_nhwc(t.contiguous(), torch.empty_like(t), taps, nt, tmp)
    bx, by = bl(x), bl(y)
    lhs, rhs = (bx.double() * y.double()).sum().item(), (x.double() * by.double()).sum().item()
    assert abs(lhs - rhs) < 1e-5 * max(1.0, abs(lhs))
    z = bl(2.0 * x + 3.0 * y)
    assert (z - (2.0 * bx + 3.0 * by)).abs().max().item() < 2e-5
    ones = bl(torch.ones(shape, device="cuda"))
    g = np.array(ops.gauss_kernel_1d(se, ks), np.float64)
    half = nt // 2
    inner = g[max(0, half - H // 2):half + (H - H // 2)].sum() ** 2              # taps that stay inside the image from its centre
    assert abs(ones[0, H // 2, W // 2, 0].item() - inner) < 2e-5
    assert abs(ones[B - 1, 0, 0, C - 1].item() - g[half:].sum() ** 2) < 2e-5
    assert torch.equal(ones[0], ones[B - 1])                                      # every image of the batch alike


def test_blur_rejects_bad_arguments():
    from blurred_gan_amd import ops
    x = torch.zeros(1, 4, 4, 3, device="cuda")
    taps = torch.ones(4, device="cuda")
    with pytest.raises(ValueError):
        ops.blur_nhwc(x, torch.empty_like(x), taps, 4)          # even tap count


@pytest.mark.parametrize("B,H,W,C,sigma", [(5, 64, 64, 3, 5.0), (3, 32, 32, 3, 3.0), (4, 28, 28, 4, 2.5), (2, 64, 48, 1, 4.0), (7, 40, 64, 3, 10.5),
                                           (6, 28, 28, 1, 0.05), (6, 28, 28, 1, 1.0), (3, 8, 8, 3, 0.9), (2, 64, 64, 3, 1.6)])
def test_three_source_blur_with_the_lerp_formed_on_the_fly(B, H, W, C, sigma):
    """bg_blur3_lerp_nhwc_f32 (wgan.py:138-139, 239-240 in one launch): [blur(f); blur(r); blur(r + a (f - r))] must be BIT-identical
    to bg_lerp_f32 followed by three bg_blur_nhwc_f32 calls -- x-hat is formed with the same expression while its rows are staged.
    Under 13 taps (the MNIST schedule: 3 taps at sigma 0.05) the single-source call runs the sliding-window kernel, whose sum order
    differs from the Toeplitz product's: there the comparison is to rounding, and against the float64 oracle."""
    from blurred_gan_amd import ops
    g = torch.Generator(device="cuda").manual_seed(B * 100 + H)
    f = torch.rand(B, H, W, C, device="cuda", generator=g) * 2 - 1
    r = torch.rand(B, H, W, C, device="cuda", generator=g) * 2 - 1
    a = torch.rand(B, device="cuda", generator=g)
    ks, se, nt = ops.blur_policy(sigma, H, W)
    taps = torch.tensor(ops.gauss_kernel_1d(se, ks), device="cuda")
    assert ops.blur3_lerp_supported(B, H, W, C, nt), (H, W, C, nt)
    y3 = ops.blur3_lerp(f, r, a, torch.empty(3 * B, H, W, C, device="cuda"), taps, nt)
    xhat = ops.lerp(r, f, a, torch.empty_like(f))
    nb = ops.blur_workspace_bytes(B, H, W, C, nt)
    tmp = torch.empty(nb // 4 + 4, device="cuda") if nb else None
    for i, src in enumerate((f, r, xhat)):
        want = ops.blur_nhwc(src, torch.empty_like(f), taps, nt, tmp)
        if nt >= 13:
            assert torch.equal(y3[i * B:(i + 1) * B], want), i
        else:
            ref = O.blur_images(src.cpu().numpy().astype(np.float64), sigma)
            np.testing.assert_allclose(y3[i * B:(i + 1) * B].cpu().numpy(), ref, rtol=0, atol=2e-6)
            np.testing.assert_allclose(y3[i * B:(i + 1) * B].cpu().numpy(), want.cpu().numpy(), rtol=0, atol=1e-6)
    # geometries of other kernels are refused, not silently run elsewhere
    assert not ops.blur3_lerp_supported(2, 30, 30, 1, 3) and not ops.blur3_lerp_supported(2, 128, 128, 3, 31)
    with pytest.raises(ValueError):
        ops.blur3_lerp(torch.zeros(B, 30, 30, 1, device="cuda"), torch.zeros(B, 30, 30, 1, device="cuda"), a,
                       torch.empty(3 * B, 30, 30, 1, device="cuda"), taps, nt)


def test_fused_critic_batch_equals_separate_launches(tmp_path, monkeypatch):
    """The whole discriminator_step with the three-source launch against BGAN_NO_FUSED_BLUR3=1 (lerp + three blurs): bit-identical
    state after two steps."""
    import blurred_gan_amd as bg
    from blurred_gan_amd import models

    def run(no_fuse):
        if no_fuse:
            monkeypatch.setenv("BGAN_NO_FUSED_BLUR3", "1")
        else:
            monkeypatch.delenv("BGAN_NO_FUSED_BLUR3", raising=False)
        bg.set_seed(77)
        gen, disc = models.DCGANGenerator(arch="celeba64"), models.DCGANDiscriminator(arch="celeba64")
        hp = bg.BlurredWGANGP.HyperParameters(initial_blur_std=5.0, global_batch_size=8, batch_size=8)
        gan = bg.BlurredWGANGP(gen, disc, hp, bg.TrainingConfig(log_dir=str(tmp_path / "log")), step_replay=False)
        g = torch.Generator().manual_seed(1)
        out = [gan.train_on_batch((torch.rand(8, 64, 64, 3, generator=g) * 2 - 1).cuda()) for _ in range(2)]
        return out, gan.discriminator.store.theta.clone(), gan.generator.store.theta.clone()
    a, b = run(False), run(True)
    assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


@pytest.mark.parametrize("shape,std", [((128, 128, 128, 3), 5.0), ((64, 256, 256, 3), 5.0), ((64, 256, 256, 3), 23.5), ((64, 256, 256, 3), 42.34),
                                       ((256, 64, 64, 3), 5.0), ((768, 64, 64, 3), 5.0)])
def test_blur_at_full_baseline_batches_matches_oracle_on_sampled_images(shape, std):
    """The FULL-SIZE launches of BASELINE.json's configurations -- C2 (256 and the critic's 3 x 256 images of 64x64), C4 (128 images of
    128x128), C5 (64 images of 256x256 at 31 / 143 / 255 taps: the fused strips and the 32-row panel kernel with its XCD-aware
    image placement) -- against the float64 oracle (gaussian_blur.py:50-132) on a SAMPLE of their images: the blur is per image, so
    images from the first, a middle and the last workgroup groups pin the whole-batch launch where the property checks above only
    relate its outputs to each other."""
    B, H, W, C = shape
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, size=shape).astype(np.float32)
    y, (ks, se, nt) = _run(x, std)
    picks = sorted({0, 1, 7, 8, B // 2 - 1, B // 2, B - 9, B - 1})
    ref = O.blur_images(x[picks].astype(np.float64), std)
    np.testing.assert_allclose(y[picks], ref, rtol=POINT_RTOL, atol=POINT_ATOL * 4)
    # ... and no image of the batch is left unwritten or shared: every image differs from its neighbour and from the input
    assert all(not np.array_equal(y[i], y[i + 1]) for i in range(0, B - 1, max(1, B // 16)))
