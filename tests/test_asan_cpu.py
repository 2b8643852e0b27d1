"""CPU box only (listed in .gpurunignore together with tools/host_sanitizer_build.py: sanitizer builds never travel to a GPU
box).  The host side of the C ABI under AddressSanitizer -- SURVEY.md section 5."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_ASAN_SCRIPT = r"""
import ctypes as C, importlib.util, os, sys
root = sys.argv[1]
spec = importlib.util.spec_from_file_location("bg_lib", os.path.join(root, "blurred-gan_amd", "_lib.py"))
L = importlib.util.module_from_spec(spec); spec.loader.exec_module(L)
lib = L.load()                                            # BGAN_HIP_LIB points at the instrumented build; every symbol binds
assert lib.bg_version() == L.ABI_VERSION
libc = C.CDLL(None); libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]; libc.free.argtypes = [C.c_void_p]
lib.bg_gauss_kernel_1d.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
ks, se, nt = C.c_float(), C.c_float(), C.c_int()
for std in (0.01, 0.05, 0.5, 1.0, 4.94, 5.0, 10.5, 23.5, 42.34, 1000.0):
    for hw in ((3, 3), (8, 8), (28, 28), (64, 64), (128, 128), (256, 256), (12, 20), (1024, 16)):
        assert lib.bg_blur_policy(std, hw[0], hw[1], C.byref(ks), C.byref(se), C.byref(nt)) == 0
        n = nt.value
        buf = libc.malloc(4 * n)                          # malloc'ed at EXACT capacity: one float past it is a heap overflow ASan reports
        got = C.c_int()                                   # (ctypes arrays live in pymalloc arenas, which the sanitizer does not fence)
        assert lib.bg_gauss_kernel_1d(se.value, ks.value, buf, n, C.byref(got)) == 0 and got.value == n
        libc.free(buf)
        if n > 1:
            small = libc.malloc(4 * (n - 1))
            assert lib.bg_gauss_kernel_1d(se.value, ks.value, small, n - 1, None) == -5
            libc.free(small)
# planners / workspace queries (pure host code): the geometries of every configuration plus ragged ones
for B in (1, 3, 64, 128, 256, 768):
    for (H, W, Ci, Co, s) in ((64, 64, 3, 32, 2), (32, 32, 32, 64, 2), (16, 16, 64, 128, 2), (8, 8, 128, 256, 2), (4, 4, 256, 512, 2),
                              (4, 4, 512, 512, 1), (128, 128, 16, 32, 2), (64, 64, 32, 3, 1), (28, 28, 1, 64, 2), (7, 9, 32, 64, 1), (5, 5, 8, 4, 2)):
        lib.bg_conv2d_bwd_filter_workspace_bytes(B, H, W, Ci, Co, 5, s)
        lib.bg_conv2d_splitk_workspace_bytes(0, B, H, W, Ci, Co, 5, s)
        lib.bg_conv2d_splitk_workspace_bytes(1, B, H, W, Ci, Co, 5, s)
    for (H, W, Cc) in ((64, 64, 3), (28, 28, 1), (128, 128, 3), (256, 256, 3), (218, 178, 3), (5, 7, 2), (130, 66, 4), (40, 24, 8)):
        for T in (3, 13, 31, 65, 143, 255):
            lib.bg_blur_workspace_bytes(B, H, W, Cc, T)
    for (M, Cn) in ((B, 8192), (B * 16, 512), (B * 4096, 32), (7, 3)):
        lib.bg_bn_workspace_bytes(M, Cn); lib.bg_colsum_workspace_bytes(M, Cn)
# argument errors come back as statuses (checked before anything touches a device)
assert lib.bg_conv2d_fwd(None, None, None, 1, 4, 4, 32, 32, 5, 1, None, None) == -6
assert lib.bg_conv2d_bwd_filter(None, None, None, 1, 4, 4, 32, 32, 5, 1, 0.0, 1.0, None, 0, None) == -6
assert lib.bg_blur_nhwc_f32(None, None, 1, 8, 8, 3, None, 3, None, None) == -6
assert lib.bg_gemm_f32(None, None, None, 1, 1, 1, 0, 0, None, 0.0, 1.0, None) == -6
assert lib.bg_blur_policy(1.0, 0, 8, C.byref(ks), C.byref(se), C.byref(nt)) == -1 and b"bg_blur_policy" in lib.bg_last_error()
h = C.c_void_p()
assert lib.bg_comm_init(C.byref(h), 2, 2, b"\0" * 128) == -1 and lib.bg_comm_destroy(None) == 0
assert lib.bg_prof_enable(0) == 0 and lib.bg_prof_reset() == 0 and lib.bg_prof_count() == 0
assert lib.bg_prof_get(0, None, 0, None, None, None) == -1
lib.bg_range_enable(0); lib.bg_range_push(b"x"); lib.bg_range_pop()
print("asan host pass ok")
"""


def test_host_side_of_the_abi_under_address_sanitizer(tmp_path):
    """the HOST side of the C ABI (argument checks, the conv / blur / filter-gradient planners, the sigma
    policy and tap generation with exact-size output buffers) built with -fsanitize=address and driven from a child python with
    the sanitizer runtime preloaded.  Host code only (--cuda-host-only: the build holds no device code and launches nothing);
    GPU-side sanitizers are not available on this pool."""
    spec = importlib.util.spec_from_file_location("host_sanitizer_build", os.path.join(ROOT, "tools", "host_sanitizer_build.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    lib, rt = tool.build()
    if not os.path.exists(rt):
        pytest.skip("clang AddressSanitizer runtime not found")
    script = tmp_path / "asan_host.py"
    script.write_text(_ASAN_SCRIPT)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", BGAN_HIP_LIB=lib)
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "asan host pass ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
