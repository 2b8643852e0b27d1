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
