#!/usr/bin/env python
"""CPU box only (listed in .gpurunignore: it never travels to a GPU box, where sanitizer builds are not allowed).

Builds the HOST side of the C ABI -- argument checks, the conv / blur / filter-gradient planners, the sigma policy and tap
generation -- with clang's AddressSanitizer, for tests/test_asan_cpu.py.  Host code ONLY
(--cuda-host-only): the objects hold no device code, nothing can be launched from this build, and the per-translation-unit
device blobs the module constructors refer to are linked in as empty stubs.  Prints the library path and the sanitizer runtime."""
import glob
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_build", "asan")
LIB = os.path.join(OUT, "libbgan_hip_asan.so")
SAN = "-fsanitize=address"
FLAGS = ["--offload-arch=gfx950", "--cuda-host-only", "-O1", "-g", "-fPIC", "-std=c++17", "-ffp-contract=off", "-w", SAN, "-shared-libasan"]


def _build_module():
    spec = importlib.util.spec_from_file_location("bgan_build", os.path.join(ROOT, "blurred-gan_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def runtime(hipcc):
    p = subprocess.run([hipcc, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(p) or not os.path.exists(p):
        c = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
        p = c[0] if c else p
    return p


def stubs(objs, run):
    """Every host object refers to the device code object of its translation unit (__hip_fatbin_<hash>), which a host-only
    compile does not produce: empty blobs keep the module constructors linkable."""
    syms = set()
    for o in objs:
        out = subprocess.run(["nm", "-u", o], capture_output=True, text=True).stdout
        syms.update(l.split()[-1] for l in out.splitlines() if "__hip_fatbin_" in l)
    stub_c, stub_o = os.path.join(OUT, "fatbin_stub.c"), os.path.join(OUT, "fatbin_stub.o")
    with open(stub_c, "w") as f:
        for sym in sorted(syms):
            f.write(f'__attribute__((section(".hip_fatbin"), aligned(4096))) const char {sym}[4096] = {{0}};\n')
    run(["gcc", "-fPIC", "-c", stub_c, "-o", stub_o])
    objs.append(stub_o)
    return [SAN, "-shared-libasan"]


def build(verbose=False):
    mod = _build_module()
    lib = mod.build_lib(verbose=verbose, flags=FLAGS, lib=LIB, odir=OUT, post_compile=stubs)
    return lib, runtime(mod.HIPCC)


if __name__ == "__main__":
    print(*build(verbose="-v" in sys.argv))
