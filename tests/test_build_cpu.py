"""CPU-only build checks on the generated gfx950 ISA (hipcc cross-compiles without a GPU).

blur_band_t_kernel keeps its four 32x32 accumulators in a[0:63] BY HAND: the MFMAs, the zeroing and the read-out are asm
statements that name those registers, and a clobber list does not RESERVE a register.  Correctness therefore rests on the
compiler never placing a value of its own in a0..a63 inside that kernel.  This test compiles csrc/blur.hip to assembly and
asserts, for every instantiation, that the only instructions naming a0..a63 are the hand-written ones
(v_mfma_f32_32x32x2_f32 on the four aligned groups, v_accvgpr_write_b32 aN, 0 and v_accvgpr_read_b32 vM, aN) and that nothing
spills."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-w"]      # = blurred-gan_amd/build.py

AGPR = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")
HAND_MFMA = re.compile(r"^v_mfma_f32_32x32x2_f32 a\[(\d+):(\d+)\], v\d+, v\d+, a\[(\d+):(\d+)\]$")
HAND_ZERO = re.compile(r"^v_accvgpr_write_b32 a(\d+|\\r), 0$")      # a\r: the body of the .irp r,0,...,63 block of band_acc_zero
HAND_READ = re.compile(r"^v_accvgpr_read_b32 v\d+, a(\d+)$")


def _low_agprs(ins):
    """AGPR indices below 64 that an instruction names."""
    out = []
    for m in AGPR.finditer(ins):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return [r for r in out if r < 64]


@pytest.fixture(scope="module")
def blur_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = tmp_path_factory.mktemp("isa") / "blur.s"
    subprocess.run([HIPCC, *FLAGS, "-S", "--cuda-device-only", os.path.join(ROOT, "blurred-gan_amd", "csrc", "blur.hip"), "-o", str(out)],
                   check=True, capture_output=True)
    return out.read_text()


def test_band_kernel_accumulator_registers_are_only_touched_by_hand(blur_isa):
    bodies = re.findall(r"^(_ZN\S*blur_band_t_kernel\S*):.*?\n(.*?)^\s*\.end_amdhsa_kernel", blur_isa, flags=re.S | re.M)
    assert len(bodies) >= 6, [b[0] for b in bodies]          # C = 1..4 x loader variants
    for name, body in bodies:
        n_mfma = n_zero = n_read = 0
        for line in body.splitlines():
            ins = line.split(";")[0].strip()
            if not ins or ins.startswith(".") or ins.endswith(":"):
                continue
            if HAND_ZERO.match(ins):
                n_zero += 64 if "\\r" in ins else 1
                continue
            low = _low_agprs(ins)
            if not low:
                continue
            m = HAND_MFMA.match(ins)
            if m:
                lo, hi, lo2, hi2 = (int(g) for g in m.groups())
                assert (lo, hi) == (lo2, hi2) and lo % 16 == 0 and hi == lo + 15 and hi < 64, (name, ins)
                n_mfma += 1
            elif HAND_READ.match(ins):
                n_read += 1
            else:
                raise AssertionError(f"{name}: compiler-generated instruction touches the hand-allocated accumulators: {ins}")
        assert n_mfma >= 16 and n_zero == 64 and n_read >= 64, (name, n_mfma, n_zero, n_read)
        assert ".irp r," + ",".join(str(i) for i in range(64)) + "\n" in body, name
        meta = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\n(.*?)\.end_amdhsa_kernel", blur_isa, flags=re.S)
        assert meta, name
    # spill counts of every band kernel, from the code-object metadata
    for m in re.finditer(r"\.name:\s+(\S*blur_band_t_kernel\S*)\n(.*?)\.wavefront_size", blur_isa, flags=re.S):
        spills = re.findall(r"\.(?:sgpr|vgpr)_spill_count:\s+(\d+)", m.group(2))
        assert spills and all(int(s) == 0 for s in spills), (m.group(1), spills)
    scratch = re.findall(r"\.set (\S*blur_band_t_kernel\S*)\.private_seg_size, (\d+)", blur_isa)
    assert scratch and all(int(v) == 0 for _, v in scratch), scratch


def _isa(tmp_path_factory, src):
    out = tmp_path_factory.mktemp("isa") / (src + ".s")
    subprocess.run([HIPCC, *FLAGS, "-S", "--cuda-device-only", os.path.join(ROOT, "blurred-gan_amd", "csrc", src), "-o", str(out)],
                   check=True, capture_output=True)
    return out.read_text()


def _resources(isa, pattern):
    """{kernel: (vgpr spills, sgpr spills, scratch bytes)} of the kernels whose mangled name contains `pattern`."""
    res = {}
    for m in re.finditer(r"\.name:\s+(\S*" + pattern + r"\S*)\n(.*?)\.wavefront_size", isa, flags=re.S):
        body = m.group(2)
        g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", body).group(1))
        res[m.group(1)] = (g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"))
    return res


def test_hot_kernels_do_not_spill(tmp_path_factory):
    """the statistics variant of the 64 x 64 gather-GEMM tile must not be forced to four waves per SIMD (it would
    spill); and the two kernels of round 4 that live on a tight register budget -- the strip-resident filter gradient (208
    accumulator registers + fragments) and the panel blur -- stay free of spills and scratch."""
    for src, pattern, least in (("conv_igemm.hip", "conv_igemm_kernel", 6), ("conv_wgrad.hip", "conv_wgrad_strip_kernel", 3),
                                ("blur_panel.hip", "blur_panel_kernel", 1)):
        isa = _isa(tmp_path_factory, src)
        res = _resources(isa, pattern)
        assert len(res) >= least, (src, list(res))
        # SGPR spills go to VGPR lanes (v_writelane), not to memory: the wide tiles carry a few; what must stay zero is VGPR spills and scratch
        assert all(v[0] == 0 and v[2] == 0 for v in res.values()), {k: v for k, v in res.items() if v[0] or v[2]}
        if pattern == "conv_wgrad_strip_kernel":
            # the k loop's order is pinned by volatile asm: between two MFMAs of a pair sits at most ONE other memory instruction
            body = re.search(r"^_ZN\S*conv_wgrad_strip_kernelILi5\S*:[^\n]*\n(.*?)s_endpgm", isa, flags=re.S | re.M).group(1)
            ops = [ln.split()[0] for ln in body.splitlines() if ln.strip() and not ln.strip().startswith((";", "."))]
            runs, cur = [], 0
            seen_mfma = False
            for op in ops:
                if op.startswith("v_mfma"):
                    if seen_mfma:
                        runs.append(cur)
                    seen_mfma, cur = True, 0
                elif op.startswith(("ds_read", "ds_write", "buffer_load")) and seen_mfma:
                    cur += 1
            assert len(runs) > 400 and sorted(runs)[int(len(runs) * 0.9)] <= 1, sorted(runs)[-20:]
