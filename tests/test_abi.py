"""CPU-side checks of the C-ABI library: it loads, and exports every symbol include/dvd_hip.h
declares (no compute calls - there is no GPU here)."""
import os
import re

from dvd_amd import lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dvd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dvd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    syms = declared_symbols()
    assert len(syms) >= 8
    raw = lib.raw()
    missing = [s for s in syms if not hasattr(raw, s)]
    assert not missing, missing


def test_python_binding_table_matches_header():
    syms = set(declared_symbols()) - {"dvd_last_error", "dvd_version", "dvd_engine_workspace_bytes"}
    assert syms == set(lib.SIGNATURES), (syms ^ set(lib.SIGNATURES))


def test_version_and_error_string():
    assert lib.version() >= 1000
    assert isinstance(lib.raw().dvd_last_error(), bytes)


def test_argument_validation_without_gpu():
    # null pointers are rejected before any HIP call
    import ctypes as C
    rc = lib.raw().dvd_unwarp_f32(None, 16, None, None, 10, 10, C.c_float(0.987), None)
    assert rc == -1 and b"null" in lib.raw().dvd_last_error()
