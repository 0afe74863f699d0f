"""CPU property tests of the temporal weight dithering (the numpy restatement the HIP kernel is bit-exact against): what the
engine relies on when it runs ONE GEMM pass on re-rounded weights instead of the (hi, lo) split's two."""
import numpy as np

from dither_ref import dither_ref
from dvd_amd import synth


def _pair(w):
    hi = w.astype(np.float16)
    return hi, (w - hi.astype(np.float32)).astype(np.float16)


def test_result_is_one_of_the_two_f16_neighbours_and_exact_weights_are_untouched():
    w = synth.uniform("dith/cpu", (1 << 14,), -0.05, 0.05, 3)
    w[::9] = np.float16(0.0123).astype(np.float32)                    # exactly representable: lo == 0
    hi, lo = _pair(w)
    target = hi.astype(np.float64) + lo.astype(np.float64)
    for step in (0, 1, 7, 49):
        out = dither_ref(hi, lo, 0, step).astype(np.float64)
        up = np.nextafter(hi, np.float16(np.inf)).astype(np.float64)
        dn = np.nextafter(hi, np.float16(-np.inf)).astype(np.float64)
        assert np.all((out == hi) | (out == up) | (out == dn))
        assert np.all(np.abs(out - target) <= np.maximum(up - hi, hi - dn))      # never further than one f16 step from W
        assert np.array_equal(out[::9], hi[::9].astype(np.float64))


def test_mean_over_steps_converges_like_one_over_steps():
    """hash(element) + step * 2^32/phi (mod 2^32) is a golden-ratio Kronecker sequence per element: the running mean of the
    re-rounded weight approaches hi + lo like log(S)/S - 5x / 24x / 97x closer than the fixed rounding after 10 / 50 / 250
    steps on the GPU; the same factors here."""
    w = synth.uniform("dith/m", (1 << 14,), -0.05, 0.05, 5)
    hi, lo = _pair(w)
    target = hi.astype(np.float64) + lo.astype(np.float64)
    fixed = np.sqrt(np.mean((hi.astype(np.float64) - target) ** 2))
    for S, factor in ((10, 4.0), (50, 15.0), (250, 60.0)):
        acc = np.zeros(w.size)
        for s in range(S):
            acc += dither_ref(hi, lo, 0, s).astype(np.float64)
        err = np.sqrt(np.mean((acc / S - target) ** 2))
        assert err * factor < fixed, (S, err, fixed)


def test_different_elements_are_decorrelated_and_steps_differ():
    w = np.full(4096, np.float32(0.0300123), dtype=np.float32)       # the SAME weight everywhere
    hi, lo = _pair(w)
    a, b = dither_ref(hi, lo, 0, 0), dither_ref(hi, lo, 0, 1)
    frac_up = float((a != hi).mean())
    want = float(abs(lo[0].astype(np.float64)) / abs(float(np.nextafter(hi[0], np.float16(np.inf) if lo[0] > 0 else np.float16(-np.inf))) - float(hi[0])))
    assert abs(frac_up - want) < 0.03, (frac_up, want)               # the hash spreads the phase over the elements
    assert (a != b).any()                                            # and the step moves it
    assert np.array_equal(dither_ref(hi, lo, 100, 3)[:-100], dither_ref(hi, lo, 0, 3)[100:])   # elem0 is an index offset
