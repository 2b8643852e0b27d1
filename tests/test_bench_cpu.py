"""bench.py pieces that need no GPU: the work-per-image constants behind `step_conv_frac`, and the freshness of the committed
counter profile behind `roofline.traffic` (an entry is only reported while the kernel sources it was measured on are unchanged)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_conv_work_per_image_matches_the_documented_mac_counts():
    import bench
    macs = {"mnist": 329_355_264, "celeba64": 1_953_816_576, "celeba128": 2_384_715_776}      # BASELINE.md section 4 / SURVEY 8d
    for arch, m in macs.items():
        assert abs(bench.CONV_GFLOP_PER_IMAGE[arch] - 2 * m / 1e9) < 5e-4, arch
    assert bench.PEAK_MFMA_F32_TFLOPS == 157.3 and bench.PEAK_HBM_GBS == 8000.0


def test_closed_form_conv_work_matches_the_documented_counts_and_a_brute_force_tap_count():
    """bench.conv_macs_per_image walks the model tables (demo_celeba.py:51-124, demo_mnist.py:48-86): its algorithmic count must be
    the documented one to the MAC, and its 'useful' count (roofline.useful: taps on the SAME zero padding excluded) must equal a
    brute-force enumeration of every (output pixel, kh, kw) triple whose source pixel lies inside the image."""
    import bench
    macs = {"mnist": 329_355_264, "celeba64": 1_953_816_576, "celeba128": 2_384_715_776}
    for arch, m in macs.items():
        assert bench.conv_macs_per_image(arch) == m, arch

    def brute(H, W, k, s):
        Ho, Wo = -(-H // s), -(-W // s)
        pt, pl = max((Ho - 1) * s + k - H, 0) // 2, max((Wo - 1) * s + k - W, 0) // 2
        n = 0
        for oy in range(Ho):
            for ox in range(Wo):
                for kh in range(k):
                    for kw in range(k):
                        n += 0 <= oy * s + kh - pt < H and 0 <= ox * s + kw - pl < W
        return n

    for H, W, s in [(4, 4, 1), (4, 4, 2), (8, 8, 2), (7, 7, 1), (14, 14, 2), (28, 28, 2), (64, 64, 2), (64, 64, 1), (5, 9, 2), (128, 128, 2)]:
        assert bench.same_live_taps(H, 5, s) * bench.same_live_taps(W, 5, s) == brute(H, W, 5, s), (H, W, s)
    assert brute(4, 4, 5, 1) / (16 * 25) == 0.49 and abs(brute(8, 8, 5, 2) / (16 * 25) - 0.7225) < 1e-12      # the 51 % / 28 % of DESIGN section 4
    # whole steps: the share of SURVEY 8d's count that is real work
    for arch, share in (("celeba64", 0.7512), ("celeba128", 0.7921), ("mnist", 0.7661)):
        assert abs(bench.conv_macs_per_image(arch, useful=True) / bench.conv_macs_per_image(arch) - share) < 1e-4, arch


def test_library_useful_flops_is_the_same_closed_form():
    """bg_conv2d_useful_flops (what every conv launch records for bg_prof_get_useful) against the Python closed form; host-only."""
    import bench
    from blurred_gan_amd import ops
    for B, H, W, Ci, Co, s in [(256, 4, 4, 512, 512, 1), (256, 8, 8, 256, 512, 2), (3, 64, 64, 3, 32, 2), (2, 28, 28, 1, 64, 2), (5, 7, 9, 16, 8, 1), (1, 6, 5, 4, 4, 2)]:
        want = 2.0 * B * Ci * Co * bench.same_live_taps(H, 5, s) * bench.same_live_taps(W, 5, s)
        assert ops.conv2d_useful_flops(B, H, W, Ci, Co, 5, s) == want
        assert want <= 2.0 * B * (-(-H // s)) * (-(-W // s)) * Ci * Co * 25
    # the layer list's per-layer counts add up to the step totals of bench.conv_macs_per_image
    for arch in ("celeba64", "celeba128", "mnist"):
        tot, first = 0.0, True
        for net, kind, H, W, Ci, Co, s in bench.conv_layers(arch):
            if kind == "dense":
                tot += (3 if net == "G" else 12) * 2.0 * Ci * Co
                continue
            mult = 4 if net == "G" else (10 if first else 12)
            first = first and net == "G"
            tot += mult * ops.conv2d_useful_flops(1, H, W, Ci, Co, 5, s)
        assert tot == 2.0 * bench.conv_macs_per_image(arch, useful=True), arch


def test_committed_traffic_profile_matches_the_kernel_sources():
    """Every entry of profiles/hbm_traffic.json carries the hash of the sources it was measured on; bench.py drops the entry (traffic:
    null) when they differ.  After editing a hashed source, refresh the entry on the GPU box: tools/pmc_step.sh <tag> [--arch ...]
    then tools/traffic_update.py (steps), tools/evidence_blur.sh (blur)."""
    import bench
    d = json.load(open(os.path.join(ROOT, bench.TRAFFIC_FILE)))
    seen = set()
    for e in d["entries"]:
        assert {"arch", "batch", "sources", "sources_sha", "kernels"} <= set(e), e.keys()
        for src in e["sources"]:
            assert os.path.exists(os.path.join(ROOT, src)), src
        for k, v in e["kernels"].items():
            assert v["hbm_bytes_per_launch"] > 0
            got = bench.hbm_traffic(k, e["arch"], e["batch"])
            assert got == round(v["hbm_bytes_per_launch"]), f"{e['arch']}/{k}: the profile is stale (sources edited after it was taken)"
        seen.add((e["arch"], e["batch"]))
    assert {("celeba64", 256), ("celeba128", 128), ("blur256", 64)} <= seen
    assert bench.hbm_traffic("conv_igemm_dgrad", "celeba64", 255) is None          # no profile for that workload


def test_bench_cli_parses_without_a_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--strong" in out.stdout and "blur256" in out.stdout
