"""numpy restatement of dvd_amd/csrc/dither.hip (dvd_dither_f16): integer hash + one correctly rounded fp32 division.  Test
infrastructure: the GPU test compares the kernel with it bit for bit, the CPU test checks the properties the engine relies on."""
import numpy as np


def dither_ref(hi16: np.ndarray, lo16: np.ndarray, elem0: int, step: int) -> np.ndarray:
    """numpy restatement of dither.hip (integer hash + one correctly rounded fp32 division): bit-exact."""
    hb = hi16.view(np.uint16).astype(np.uint32).ravel()
    lb = lo16.view(np.uint16).astype(np.uint32).ravel()
    with np.errstate(over="ignore"):
        g = (np.uint32(elem0) + np.arange(hb.size, dtype=np.uint32)).astype(np.uint32)
        h = (g * np.uint32(0x9E3779B1)).astype(np.uint32)
        h ^= h >> np.uint32(15)
        h = (h * np.uint32(0x85EBCA77)).astype(np.uint32)
        h ^= h >> np.uint32(13)
        h = (h * np.uint32(0xC2B2AE3D)).astype(np.uint32)
        h ^= h >> np.uint32(16)
        u = (h + np.uint32((step * 0x9E3779B9) & 0xFFFFFFFF)).astype(np.uint32)
    lo_zero = (lb & 0x7FFF) == 0
    hi_zero = (hb & 0x7FFF) == 0
    opposite = ((hb ^ lb) & 0x8000) != 0
    nb = np.where(hi_zero, (lb & 0x8000) | 1, np.where(opposite, hb - 1, hb + 1)) & 0xFFFF
    is_inf = (nb & 0x7C00) == 0x7C00
    f = lambda b: b.astype(np.uint16).view(np.float16).astype(np.float32)  # noqa: E731
    hf, lf, nf = f(hb), f(lb), f(nb)
    with np.errstate(divide="ignore", invalid="ignore"):
        frac = np.abs(lf) / np.abs(nf - hf)
    frac = np.minimum(np.nan_to_num(frac, nan=0.0, posinf=0.99999994), np.float32(0.99999994)).astype(np.float32)
    thr = (frac * np.float32(4294967296.0)).astype(np.uint64).astype(np.uint32)
    out = np.where(lo_zero | is_inf, hb, np.where(u < thr, nb, hb))
    return out.astype(np.uint16).view(np.float16).reshape(hi16.shape)
