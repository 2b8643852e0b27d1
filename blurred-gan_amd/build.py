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
SOURCES = ["runtime.hip", "blur.hip", "blur_panel.hip", "conv_igemm.hip", "conv_rows.hip", "conv_c16.hip", "conv_wgrad.hip", "misc.hip", "comm.hip", "program.hip"]
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


def build_lib(force=False, verbose=True, flags=None, lib=None, odir=None, post_compile=None):
    """flags / lib / odir: an alternative build of the same sources (tools/host_sanitizer_build.py: the host-only build the CPU
    tests run under a sanitizer); post_compile(objs) may append objects before the link."""
    objs, jobs = [], []
    hdr_time = max(os.path.getmtime(h) for h in _deps())
    flags, lib, odir = flags or FLAGS, lib or LIB, odir or CSRC
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
    link_extra = []
    if post_compile is not None and (jobs or not os.path.exists(lib)):
        link_extra = post_compile(objs, run)
    if jobs or not os.path.exists(lib) or any(_newer(o, lib) for o in objs):      # objects newer than the library: a build that died before the link
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *link_extra, *objs, "-ldl", "-o", lib])
    return lib


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
