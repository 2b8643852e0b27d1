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
