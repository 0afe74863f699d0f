"""ctypes binding of libdvd_hip.so (include/dvd_hip.h).

The HIP library IS the product: there is no CPU or eager-PyTorch fallback.  Importing this
module loads `dvd_amd/libdvd_hip.so` (built in-tree by `__graft_entry__.build()` /
`make -C dvd_amd/csrc`) and raises if it is missing; every wrapper raises `DvdError` with
the library's message on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
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


def _sig(name, argtypes, restype=C.c_int):
    fn = getattr(_lib, name)
    fn.argtypes = argtypes
    fn.restype = restype
    return fn


_lib.dvd_last_error.restype = C.c_char_p
_lib.dvd_version.restype = C.c_int

# name -> argtypes; kept in one table so tests can check every symbol of include/dvd_hip.h
SIGNATURES = {
    "dvd_grid_sample_bilinear_zeros_ac": [c_void, c_void, c_void, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, c_void],
    "dvd_unwarp_f32": [c_void, C.c_int, c_void, c_void, C.c_int, C.c_int, C.c_float, c_void],
    "dvd_unwarp_u8": [c_void, C.c_int, c_void, c_void, C.c_int, C.c_int, C.c_float, c_void],
    "dvd_unwarp_grid": [c_void, C.c_int, c_void, C.c_int, C.c_int, C.c_float, c_void],
    "dvd_sched_step": [C.POINTER(SchedCoef), c_void, c_void, c_void, c_void, c_void, C.c_int, C.c_int, c_void],
    "dvd_hyp_mean_clamp": [c_void, c_void, C.c_int, C.c_int, C.c_int, c_void],
    "dvd_selftest_mfma": [c_void, c_void, c_void, c_void, c_void],
}


def _bind_all():
    for name, args in SIGNATURES.items():
        _sig(name, args)


_bind_all()


def call(name: str, *args):
    """Invoke a status-returning entry point; raise DvdError on failure."""
    rc = getattr(_lib, name)(*args)
    if rc != 0:
        raise DvdError(f"{name} failed ({rc}): {_lib.dvd_last_error().decode()}")


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
