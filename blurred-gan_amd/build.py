"""Builds libbgan_hip.so (hand-written HIP kernels for gfx950 + the C ABI of include/bgan.h) in-tree.

hipcc cross-compiles for gfx950 without a GPU; the .so travels to the GPU box with the snapshot.
Usage: python blurred-gan_amd/build.py [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbgan_hip.so")
SOURCES = ["runtime.hip", "blur.hip", "conv_igemm.hip", "conv_rows.hip", "conv_c16.hip", "conv_wgrad.hip", "misc.hip", "comm.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]   # fp contraction off: fmaf() is explicit where wanted
FLAGS += os.environ.get("BG_EXTRA_FLAGS", "").split()


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "bgan.h"))
    return hdrs


ASAN_DIR = os.path.join(os.path.dirname(HERE), "tools", "_build", "asan")
ASAN_LIB = os.path.join(ASAN_DIR, "libbgan_hip_asan.so")
# HOST code only (--cuda-host-only: no device code objects, so nothing can be launched from this build -- it exists for the
# argument checks, planners and policy maths), instrumented; seconds to compile
ASAN_FLAGS = ["--offload-arch=gfx950", "--cuda-host-only", "-O1", "-g", "-fPIC", "-std=c++17", "-ffp-contract=off", "-w",
              "-fsanitize=address", "-shared-libasan"]


def asan_runtime():
    """Path of clang's shared AddressSanitizer runtime (to LD_PRELOAD into the python that dlopens the instrumented library)."""
    r = subprocess.run([HIPCC, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    p = r.stdout.strip()
    if not os.path.isabs(p):
        import glob
        c = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
        p = c[0] if c else p
    return p


def build_lib(force=False, verbose=True, asan=False):
    """asan=True: the AddressSanitizer build of the HOST side of the C ABI (argument checks, planners, policy maths) for the
    CPU tests -- never run on the GPU box."""
    objs, jobs = [], []
    hdr_time = max(os.path.getmtime(h) for h in _deps())
    flags, lib, odir = (ASAN_FLAGS, ASAN_LIB, ASAN_DIR) if asan else (FLAGS, LIB, CSRC)
    os.makedirs(odir, exist_ok=True)
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(odir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(src, obj) or hdr_time > os.path.getmtime(obj):
            jobs.append([HIPCC, *flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if asan and (jobs or not os.path.exists(lib)):
        # every host object refers to the device code object of its translation unit (__hip_fatbin_<hash>), which a host-only
        # compile does not produce: empty blobs keep the module constructors linkable; nothing is ever launched from this build
        syms = set()
        for o in objs:
            out = subprocess.run(["nm", "-u", o], capture_output=True, text=True).stdout
            syms.update(l.split()[-1] for l in out.splitlines() if "__hip_fatbin_" in l)
        stub_c, stub_o = os.path.join(odir, "fatbin_stub.c"), os.path.join(odir, "fatbin_stub.o")
        with open(stub_c, "w") as f:
            for sym in sorted(syms):
                f.write(f'__attribute__((section(".hip_fatbin"), aligned(4096))) const char {sym}[4096] = {{0}};\n')
        run(["gcc", "-fPIC", "-c", stub_c, "-o", stub_o])
        objs.append(stub_o)
    if jobs or not os.path.exists(lib):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *(["-fsanitize=address", "-shared-libasan"] if asan else []), *objs, "-ldl", "-o", lib])
    return lib


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, asan="--asan" in sys.argv))
