"""ctypes binding of libdvd_hip.so (include/dvd_hip.h).

The HIP library IS the product: there is no CPU or eager-PyTorch fallback.  Importing this
module loads `dvd_amd/libdvd_hip.so` (built in-tree by `__graft_entry__.build()` /
`make -C dvd_amd/csrc`) and raises if it is missing; every wrapper raises `DvdError` with
the library's message on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os

# torch must be imported BEFORE the library is loaded: the PyTorch-ROCm wheel bundles its own
# libamdhip64.so.7 and libdvd_hip.so must bind to that same HIP runtime (same SONAME => the loader
# reuses it); loading ours first would create a second runtime that sees no device.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product loads the in-tree build and nothing else: no environment variable can put another library under it.
# benchmarks/ and the `lab` pytest fixture swap in the lab build explicitly, in their own process (use_library below).
LIB_PATH = os.path.join(_HERE, "libdvd_hip.so")


class DvdError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise DvdError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C dvd_amd/csrc`.  The DvD engine has no non-HIP fallback.")
    return C.CDLL(LIB_PATH)


_lib = _load()

c_f32p = C.POINTER(C.c_float)
c_void = C.c_void_p


class SchedCoef(C.Structure):
    _fields_ = [("kind", C.c_int), ("c_recip", C.c_float), ("c_recipm1", C.c_float),
                ("sqrt_abar_prev", C.c_float), ("dir_coef", C.c_float), ("coef1", C.c_float),
                ("coef2", C.c_float), ("sigma", C.c_float)]


class GemmDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("batch", C.c_int),
                ("A", C.c_void_p), ("lda", C.c_int), ("strideA", C.c_long),
                ("B", C.c_void_p), ("ldb", C.c_int), ("strideB", C.c_long),
                ("B_lo", C.c_void_p), ("A_lo", C.c_void_p), ("lo_scale", C.c_float),
                ("C32", C.c_void_p), ("ldc", C.c_int), ("strideC32", C.c_long),
                ("C16", C.c_void_p), ("ldc16", C.c_int), ("strideC16", C.c_long),
                ("bias", C.c_void_p), ("bias_row", C.c_int), ("strideBias", C.c_long),
                ("act", C.c_int),
                ("pos", C.c_void_p), ("ldpos", C.c_int), ("pos_rows", C.c_int),
                ("gate", C.c_void_p), ("ldgate", C.c_int), ("gate_rows", C.c_int), ("strideGate", C.c_long),
                ("res", C.c_void_p), ("ldres", C.c_int), ("strideRes", C.c_long), ("small_tiles", C.c_int)]


class AttnDesc(C.Structure):
    _fields_ = [("head_dim", C.c_int), ("heads", C.c_int), ("batch", C.c_int), ("tq", C.c_int), ("tk", C.c_int),
                ("kv_batch_div", C.c_int),
                ("Q", C.c_void_p), ("ldq", C.c_int), ("strideQ", C.c_long),
                ("K", C.c_void_p), ("ldk", C.c_int), ("strideK", C.c_long),
                ("Vt", C.c_void_p), ("ldvt", C.c_int), ("strideVt", C.c_long),
                ("O", C.c_void_p), ("ldo", C.c_int), ("strideO", C.c_long),
                ("scale", C.c_float)]


class CnOp(C.Structure):
    _fields_ = [("op", C.c_int), ("a", C.c_int), ("b", C.c_int), ("dst", C.c_int), ("ks", C.c_int), ("dil", C.c_int),
                ("cout", C.c_int), ("act", C.c_int), ("w_off", C.c_long), ("h", C.c_int), ("w", C.c_int),
                ("flag", C.c_int)]


NON_STATUS = {"dvd_last_error", "dvd_version", "dvd_engine_workspace_bytes", "dvd_engine_tensor_count",
              "dvd_flash_attn_kernel_name", "dvd_convnet_workspace_bytes", "dvd_convnet_weight_floats",
              "dvd_ingest_scratch_bytes"}

# name -> argtypes; kept in one table so tests can check every symbol of include/dvd_hip.h
SIGNATURES = {
    "dvd_grid_sample_bilinear_zeros_ac": [c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, c_void],
    "dvd_unwarp_f32": [c_void, C.c_int, c_void, c_void, C.c_int, C.c_int, C.c_float, c_void],
    "dvd_unwarp_u8": [c_void, C.c_int, c_void, c_void, C.c_int, C.c_int, C.c_float, c_void],
    "dvd_unwarp_f32_batch": [c_void, C.c_int, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_float, c_void],
    "dvd_unwarp_u8_batch": [c_void, C.c_int, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_float, c_void],
    "dvd_unwarp_grid": [c_void, C.c_int, c_void, C.c_int, C.c_int, C.c_float, c_void],
    "dvd_sched_step": [C.POINTER(SchedCoef), c_void, c_void, c_void, c_void, c_void, C.c_int, C.c_int, c_void],
    "dvd_hyp_mean_clamp": [c_void, c_void, C.c_int, C.c_int, C.c_int, c_void],
    "dvd_selftest_mfma": [c_void, c_void, c_void, c_void, c_void],
    "dvd_gemm_nt": [C.POINTER(GemmDesc), c_void],
    "dvd_flash_attn": [C.POINTER(AttnDesc), c_void],
    "dvd_embed_obs_ln": [c_void, c_void, c_void, c_void, c_void, c_void, C.c_int, C.c_int, c_void],
    "dvd_layernorm_rows": [c_void, C.c_int, C.c_long, c_void, C.c_int, C.c_long, C.c_int, C.c_long, C.c_int, c_void,
                           c_void, c_void, c_void, C.c_int, C.c_int, C.c_float, c_void],
    "dvd_build_r_rows": [c_void, c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void],
    "dvd_patch_rows": [c_void, C.c_long, C.c_long, C.c_long, C.c_long, c_void, C.c_int, C.c_int, C.c_int, C.c_int,
                       c_void],
    "dvd_dwconv3x3": [c_void, c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, c_void],
    "dvd_colmean": [c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, c_void],
    "dvd_posenc_add": [c_void, c_void, c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, c_void],
    "dvd_small_linear": [c_void, C.c_int, c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                         C.c_int, c_void],
    "dvd_final_tokens": [c_void, c_void, c_void, c_void, c_void, C.c_int, C.c_int, c_void, c_void, c_void, c_void,
                         c_void, C.c_int, C.c_int, c_void],
    "dvd_im2col3x3": [c_void, C.c_long, C.c_long, C.c_long, c_void, C.c_int, C.c_int, C.c_int, C.c_int, c_void],
    "dvd_conv3x3_nhwc": [c_void, C.c_int, c_void, C.c_int, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, c_void],
    "dvd_maxpool2_nhwc": [c_void, c_void, C.c_int, C.c_int, C.c_int, c_void],
    "dvd_resize_bilinear_nhwc": [c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void],
    "dvd_nhwc_to_nchw": [c_void, c_void, C.c_int, C.c_int, C.c_int, c_void],
    "dvd_convnet_create": [C.POINTER(CnOp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(c_void)],
    "dvd_convnet_create_batched": [C.POINTER(CnOp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(c_void)],
    "dvd_convnet_destroy": [c_void],
    "dvd_convnet_slot_shape": [c_void, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "dvd_convnet_run": [c_void, c_void, c_void, c_void, C.c_long, C.c_int, C.POINTER(C.c_int), C.POINTER(c_void), c_void],
    "dvd_resize_bilinear_nchw": [c_void, c_void, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void],
    "dvd_threshold_mask_mul": [c_void, c_void, c_void, c_void, C.c_int, C.c_long, C.c_float, c_void],
    "dvd_threshold_mask_mul_batch": [c_void, c_void, c_void, c_void, C.c_int, C.c_int, C.c_long, C.c_float, c_void],
    "dvd_ingest_u8": [c_void, C.c_int, C.c_int, C.c_int, c_void, C.c_int, c_void, c_void, c_void],
    "dvd_dither_f16": [c_void, c_void, c_void, C.c_long, C.c_uint, C.c_uint, c_void],
    "dvd_engine_create": [C.c_int, C.c_int, C.c_int, C.POINTER(c_void)],
    "dvd_engine_destroy": [c_void],
    "dvd_engine_bind_workspace": [c_void, c_void, C.c_long],
    "dvd_engine_tensor_count": [c_void],
    "dvd_engine_tensor_info": [c_void, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_long)],
    "dvd_engine_set_tensor": [c_void, C.c_char_p, c_void, C.c_long],
    "dvd_engine_prepare_docs": [c_void, c_void, c_void, c_void, c_void, c_void],
    "dvd_engine_feat_nchw": [c_void, c_void, c_void],
    "dvd_engine_denoise_step": [c_void, c_void, C.c_float, C.c_int, c_void, c_void, c_void, c_void],
    "dvd_engine_debug_buffer": [c_void, C.c_char_p, C.POINTER(c_void), C.POINTER(C.c_long)],
    "dvd_engine_debug_stop": [c_void, C.c_int],
    "dvd_engine_set_option": [c_void, C.c_char_p, C.c_int],
    "dvd_engine_profile": [c_void, C.c_int],
    "dvd_engine_profile_read": [c_void, C.POINTER(C.c_int), C.POINTER(C.c_double)],
}


# entry points that exist only in the lab build (benchmarks/lab/dvd_hip_lab.h; loaded through use_library)
LAB_SIGNATURES = {"dvd_gemm_debug_stamps": [c_void], "dvd_attn_debug_stamps": [c_void]}


def bind(cdll):
    """Attach argtypes / restypes to a loaded libdvd_hip.so (the product library, or the lab build in tests/benchmarks)."""
    cdll.dvd_last_error.restype = C.c_char_p
    cdll.dvd_version.restype = C.c_int
    cdll.dvd_engine_workspace_bytes.restype = C.c_long
    cdll.dvd_engine_workspace_bytes.argtypes = [C.c_void_p]
    for fn in (cdll.dvd_convnet_workspace_bytes, cdll.dvd_convnet_weight_floats):
        fn.restype, fn.argtypes = C.c_long, [C.c_void_p]
    cdll.dvd_ingest_scratch_bytes.restype, cdll.dvd_ingest_scratch_bytes.argtypes = C.c_long, [C.c_int]
    cdll.dvd_flash_attn_kernel_name.restype = C.c_char_p
    cdll.dvd_flash_attn_kernel_name.argtypes = [C.c_int, C.c_int, C.c_int]
    for name, args in SIGNATURES.items():
        fn = getattr(cdll, name)
        fn.argtypes, fn.restype = args, C.c_int
    for name, args in LAB_SIGNATURES.items():
        if hasattr(cdll, name):
            fn = getattr(cdll, name)
            fn.argtypes, fn.restype = args, C.c_int
    return cdll


bind(_lib)


def use_library(path: str):
    """Route this process through another build of the SAME library (benchmarks/_lab.py and the `lab` pytest fixture
    load benchmarks/lab/libdvd_hip_lab.so this way).  An explicit call in the caller's code - never the environment."""
    global _lib
    _lib = bind(C.CDLL(os.path.abspath(path)))
    return _lib


def call(name: str, *args):
    """Invoke a status-returning entry point; raise DvdError on failure."""
    rc = getattr(_lib, name)(*args)
    if rc != 0:
        raise DvdError(f"{name} failed ({rc}): {_lib.dvd_last_error().decode()}")


def flash_attn_kernel_name(head_dim: int, tq: int, tk: int) -> str:
    return _lib.dvd_flash_attn_kernel_name(head_dim, tq, tk).decode()


def version() -> int:
    return int(_lib.dvd_version())


def raw():
    return _lib


def ptr(t):
    """Device (or host) address of a torch tensor / None."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
