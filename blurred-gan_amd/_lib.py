"""ctypes binding of libbgan_hip.so (include/bgan.h).  No CPU fallback: a missing library or a
non-zero status raises."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BGAN_HIP_LIB") or os.path.join(HERE, "libbgan_hip.so")   # override: kernel experiments only


class BgError(RuntimeError):
    pass


class Epilogue(C.Structure):
    _fields_ = [("mode", C.c_int), ("bias", C.c_void_p), ("ref", C.c_void_p), ("keep", C.c_void_p),
                ("alpha", C.c_float), ("scale", C.c_float), ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_size_t),
                ("keep_elems", C.c_size_t), ("stats", C.c_void_p), ("stats_capacity", C.c_size_t), ("stats_rows", C.POINTER(C.c_int))]


ABI_VERSION = 5           # include/bgan.h BG_ABI_VERSION
EPI_NONE, EPI_BIAS_LRELU, EPI_MUL_GRAD, EPI_TANH, EPI_AFFINE_LRELU = 0, 1, 2, 3, 4

_p, _i, _f, _z, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_uint64

# name -> (restype, argtypes); every symbol include/bgan.h declares
SIGNATURES = {
    "bg_version": (_i, []),
    "bg_last_error": (C.c_char_p, []),
    "bg_status_string": (C.c_char_p, [_i]),
    "bg_prof_enable": (_i, [_i]),
    "bg_prof_reset": (_i, []),
    "bg_prof_count": (_i, []),
    "bg_prof_get": (_i, [_i, C.c_char_p, _i, C.POINTER(_f), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "bg_prof_get_exec": (_i, [_i, C.POINTER(C.c_double)]),
    "bg_prof_get_useful": (_i, [_i, C.POINTER(C.c_double)]),
    "bg_conv2d_useful_flops": (C.c_double, [_i] * 7),
    "bg_range_enable": (_i, [_i]),
    "bg_range_push": (_i, [C.c_char_p]),
    "bg_range_pop": (_i, []),
    "bg_blur_policy": (_i, [_f, _i, _i, C.POINTER(_f), C.POINTER(_f), C.POINTER(_i)]),
    "bg_gauss_kernel_1d": (_i, [_f, _f, C.POINTER(_f), _i, C.POINTER(_i)]),
    "bg_blur_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "bg_blur_nhwc_f32": (_i, [_p, _p, _i, _i, _i, _i, _p, _i, _p, _p]),
    "bg_blur3_lerp_supported": (_i, [_i, _i, _i, _i, _i]),
    "bg_blur3_lerp_nhwc_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p]),
    "bg_conv2d_splitk_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "bg_conv2d_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, C.POINTER(Epilogue), _p]),
    "bg_conv2d_bwd_data": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, C.POINTER(Epilogue), _p]),
    "bg_conv2d_bwd_filter_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i, _i]),
    "bg_conv2d_bwd_filter": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _f, _p, _z, _p]),
    "bg_transpose_last2": (_i, [_p, _p, _i, _i, _i, _p]),
    "bg_transpose_last2_batched": (_i, [_p, _p, _p, _i, _i, _p]),
    "bg_gemm_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _f, _f, _p]),
    "bg_colsum_workspace_bytes": (_z, [_i, _i]),
    "bg_colsum_f32": (_i, [_p, _p, _i, _i, _i, _f, _f, _p, _z, _p]),
    "bg_bn_workspace_bytes": (_z, [_i, _i]),
    "bg_bn_train_fwd": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _f, _f, _i, _f, _p, _z, _p]),
    "bg_bn_train_fwd_partials": (_i, [_p, _i, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _f, _f, _i, _f, _p]),
    "bg_bn_sums_from_partials": (_i, [_p, _i, _i, _p, _p]),
    "bg_bn_infer_fwd": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _f, _f, _p]),
    "bg_bn_train_bwd": (_i, [_p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _f, _p, _z, _p]),
    "bg_bn_fold_f32": (_i, [_p, _p, _p, _p, _f, _i, _p, _p, _p]),
    "bg_bn_fold_many_f32": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "bg_bn_stats_f32": (_i, [_p, _i, _i, _p, _p, _z, _p]),
    "bg_bn_finalize_f32": (_i, [_p, _i, _i, _p, _p, _p, _p, _f, _f, _i, _p]),
    "bg_bn_apply_f32": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _f, _p]),
    "bg_bn_finalize_apply_f32": (_i, [_p, _i, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _f, _f, _i, _f, _p]),
    "bg_bn_bwd_stats_f32": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _p, _f, _p, _p, _z, _p]),
    "bg_bn_bwd_apply_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p]),
    "bg_bn_param_grads_f32": (_i, [_p, _i, _f, _p, _p, _p]),
    "bg_lerp_f32": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "bg_row_norm_f32": (_i, [_p, _p, _i, _i, _p]),
    "bg_gp_seed_f32": (_i, [_p, _p, _f, _p, _i, _i, _p]),
    "bg_gp_seed_guarded_f32": (_i, [_p, _p, _f, _p, _i, _i, _p]),
    "bg_mul_grad_f32": (_i, [_p, _p, _p, _f, _f, _p, _z, _p]),
    "bg_tanh_bwd_f32": (_i, [_p, _p, _p, _z, _p]),
    "bg_outer_f32": (_i, [_p, _p, _p, _i, _i, _p]),
    "bg_fill_f32": (_i, [_p, _f, _z, _p]),
    "bg_scale_f32": (_i, [_p, _f, _z, _p]),
    "bg_copy_f32": (_i, [_p, _p, _z, _p]),
    "bg_wgangp_d_loss": (_i, [_p, _p, _p, _i, _f, _f, _f, _f, _p, _p, _p, _p]),
    "bg_wgan_g_loss": (_i, [_p, _i, _f, _p, _p, _p]),
    "bg_u8_normalize_resize_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "bg_adam_f32": (_i, [_p, _p, _p, _p, _z, _f, _f, _f, _f, _p]),
    "bg_uniform_f32": (_i, [_p, _z, _u64, _u64, _p]),
    "bg_keep_mask_u8": (_i, [_p, _z, _f, _u64, _u64, _p]),
    "bg_program_create": (_i, [C.POINTER(_p), _i]),
    "bg_program_destroy": (_i, [_p]),
    "bg_program_record_begin": (_i, [_p]),
    "bg_program_record_end": (_i, [_p]),
    "bg_program_size": (_i, [_p]),
    "bg_program_launches": (_i, [_p]),
    "bg_program_binds": (_i, [_p]),
    "bg_program_bind_next": (_i, [_i, _i]),
    "bg_program_slots_f64": (_p, [_p]),
    "bg_program_slots_u64": (_p, [_p]),
    "bg_program_replay": (_i, [_p, _i, _i, _p]),
    "bg_program_graph_launch": (_i, [_p, _i, _i, _p]),
    "bg_dstep": (_i, [_p, _p]),
    "bg_gstep": (_i, [_p, _p]),
    "bg_comm_unique_id": (_i, [C.c_char_p]),
    "bg_comm_init": (_i, [C.POINTER(_p), _i, _i, C.c_char_p]),
    "bg_allreduce_sum_f32": (_i, [_p, _p, _z, _p]),
    "bg_comm_query": (_i, [_p, C.POINTER(_i), C.POINTER(_i)]),
    "bg_comm_destroy": (_i, [_p]),
}

COMM_ID_BYTES = 128
BIND_ADAM_LR, BIND_RNG_OFFSET = 1, 2      # include/bgan.h BG_BIND_*

_lib = None


def load():
    """Loads the shared library and binds every exported symbol.  Raises BgError when the library
    has not been built (python blurred-gan_amd/build.py) -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BgError(f"{LIB_PATH} not found: build it with `python blurred-gan_amd/build.py` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.bg_version() != ABI_VERSION:
        raise BgError(f"ABI version mismatch: library {lib.bg_version()}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        lib = load()
        msg = lib.bg_last_error().decode("utf-8", "replace")
        kind = lib.bg_status_string(status).decode()
        exc = ValueError if status in (-1, -2, -3, -6) else BgError     # -4 HIP, -5 workspace, -7 RCCL
        raise exc(f"{what}: {kind}: {msg}")
