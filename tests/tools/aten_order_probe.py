#!/usr/bin/env python3
"""(Checker-side tool.)  How the arithmetic ORDER of the two torch-CPU kernels on the unwarp tail was found - the reference
runs F.interpolate(bilinear, align_corners=True) and F.grid_sample(bilinear, zeros, align_corners=True) on the CPU
(train_settings/dvd/evaluation.py:301-306, datasets/utils/warping.py:73), and a byte-exact u8 output needs their exact
sequence of roundings, FMA contractions of the CPU build included.  Every candidate order is evaluated in numpy float32
(an FMA = one rounding of the exact float64 product-sum) and compared BIT FOR BIT with torch; the candidates that match
(zero mismatches on every shape) are what dvd_amd/csrc/warp.hip implements and what oracle/aten_order.py restates:

  interpolate:  t_row = fma(left, lx0, right * lx1);  out = fma(t_upper, ly0, t_lower * ly1)
                src = scale * dst, scale = f32(in - 1) / f32(out - 1), i0 = min(int(src), in - 1), l1 = src - i0, l0 = 1 - l1
  grid_sample:  x = (gx + 1) * ((W - 1) / 2);  w = x - floor(x), e = (floor(x) + 1) - x  (same in y: n, s)
                out = fma(v_se, n * w, fma(v_sw, n * e, fma(v_ne, s * w, v_nw * (s * e))))

usage: python tests/tools/aten_order_probe.py      (CPU only, ~1 minute)"""
import itertools
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def mul(a, b):
    return (a * b).astype(f32)


def add(a, b):
    return (a + b).astype(f32)


def comb(p, wp, q, wq, kind):
    if kind == "nofma":
        return add(mul(p, wp), mul(q, wq))
    if kind == "fma_first":                      # fma(p, wp, q * wq)
        return fma(p, wp, mul(q, wq))
    return fma(q, wq, mul(p, wp))                # "fma_second"


def interp(x, H, W, inner, outer, order):
    h, w = x.shape
    sh, sw = f32(h - 1) / f32(H - 1), f32(w - 1) / f32(W - 1)
    ys, xs = (sh * np.arange(H, dtype=f32)).astype(f32), (sw * np.arange(W, dtype=f32)).astype(f32)
    y0, x0 = np.minimum(ys.astype(np.int64), h - 1), np.minimum(xs.astype(np.int64), w - 1)
    y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
    ly1, lx1 = (ys - y0.astype(f32)).astype(f32), (xs - x0.astype(f32)).astype(f32)
    ly0, lx0 = (f32(1) - ly1).astype(f32), (f32(1) - lx1).astype(f32)
    a, b, c, d = x[y0][:, x0], x[y0][:, x1], x[y1][:, x0], x[y1][:, x1]
    LX0, LX1 = np.broadcast_to(lx0[None], a.shape), np.broadcast_to(lx1[None], a.shape)
    LY0, LY1 = np.broadcast_to(ly0[:, None], a.shape), np.broadcast_to(ly1[:, None], a.shape)
    if order == "w then h":
        return comb(comb(a, LX0, b, LX1, inner), LY0, comb(c, LX0, d, LX1, inner), LY1, outer)
    return comb(comb(a, LY0, c, LY1, inner), LX0, comb(b, LY0, d, LY1, inner), LX1, outer)


def main():
    torch.set_num_threads(1)
    rng = np.random.RandomState(0)
    print("F.interpolate(bilinear, align_corners=True): mismatching elements per candidate order")
    kinds = ("nofma", "fma_first", "fma_second")
    for (h, w, H, W) in [(16, 16, 97, 131), (64, 64, 1873, 1353)]:
        x = rng.randn(h, w).astype(f32)
        ref = F.interpolate(torch.from_numpy(x)[None, None], size=(H, W), mode="bilinear", align_corners=True)[0, 0].numpy()
        for order, inner, outer in itertools.product(("w then h", "h then w"), kinds, kinds):
            bad = int((interp(x, H, W, inner, outer, order) != ref).sum())
            print(f"  {h}x{w} -> {H}x{W}  {order:9s} inner {inner:10s} outer {outer:10s}: {bad}{'   <== torch' if bad == 0 else ''}")
    print("F.grid_sample(bilinear, zeros, align_corners=True): see oracle/aten_order.py for the matching order; "
          "tests/test_oracle_golden.py::test_aten_order_restatement pins both against torch on every CPU run")


if __name__ == "__main__":
    main()
