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
    syms = set(declared_symbols()) - {"dvd_last_error", "dvd_version", "dvd_engine_workspace_bytes",
                                      "dvd_flash_attn_kernel_name", "dvd_convnet_workspace_bytes",
                                      "dvd_convnet_weight_floats", "dvd_ingest_scratch_bytes"}
    assert syms == set(lib.SIGNATURES), (syms ^ set(lib.SIGNATURES))


def test_product_library_reads_no_environment_and_ships_no_experiments():
    """The product .so has no getenv import and none of the lab's experiment kernels / switch names; the lab build
    (benchmarks/lab/libdvd_hip_lab.so, -DDVD_LAB) is where those live."""
    import subprocess
    so = os.path.join(ROOT, "dvd_amd", "libdvd_hip.so")
    dyn = subprocess.run(["nm", "-D", so], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in dyn
    blob = open(so, "rb").read()
    for needle in (b"DVD_ATTN_", b"DVD_GEMM_", b"DVD_WARP_", b"DVD_DWCONV_", b"flash_attn_pipe_kernel", b"flash_attn_dsplit_kernel",
                   b"flash_attn_glds64x2_kernel", b"gemm_nt_big16_kernel", b"DVD_HIP_LIB"):
        assert needle not in blob, needle
    assert "dvd_attn_debug_stamps" not in dyn and "dvd_gemm_debug_stamps" not in dyn


def test_attention_kernel_choice_depends_on_shape_only():
    assert lib.flash_attn_kernel_name(256, 20736, 20736) == "flash_attn_r64p_kernel<0>"
    assert lib.flash_attn_kernel_name(256, 1024, 1024) == "flash_attn_glds_kernel<256, 0>"
    assert lib.flash_attn_kernel_name(256, 1296, 1296) == "flash_attn_kernel<256>"
    assert lib.flash_attn_kernel_name(64, 20736, 20736) == "flash_attn_glds_kernel<64, 0>"


def test_version_and_error_string():
    assert lib.version() >= 1000
    assert isinstance(lib.raw().dvd_last_error(), bytes)


def test_argument_validation_without_gpu():
    # null pointers are rejected before any HIP call
    import ctypes as C
    rc = lib.raw().dvd_unwarp_f32(None, 16, None, None, 10, 10, C.c_float(0.987), None)
    assert rc == -1 and b"null" in lib.raw().dvd_last_error()


def test_header_is_plain_c():
    """The boundary is a C ABI: include/dvd_hip.h (and the lab header) compile as C99 and as C++ with no torch / HIP types."""
    import subprocess
    for hdr in ("include/dvd_hip.h", "benchmarks/lab/dvd_hip_lab.h"):
        path = os.path.join(ROOT, hdr)
        subprocess.run(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", path], check=True)
        subprocess.run(["g++", "-fsyntax-only", "-x", "c++", path], check=True)
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "dvd_hip.h")).read(), flags=re.S)   # code only
    assert "torch" not in text.lower() and "hipStream_t" not in text and "#include <hip" not in text
