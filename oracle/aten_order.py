"""TEST INFRASTRUCTURE (checker only - never imported by dvd_amd/): a numpy float32 restatement of the exact arithmetic
ORDER of the two torch-CPU kernels the reference's unwarp tail runs, i.e. of what dvd_amd/csrc/warp.hip implements step by
step since round 5:

  interpolate_ac   F.interpolate(x, size, mode='bilinear', align_corners=True)      (train_settings/dvd/evaluation.py:301,304)
  grid_sample_ac   F.grid_sample(src, grid, 'bilinear', 'zeros', align_corners=True) (datasets/utils/warping.py:73)
  unwarp_tail      evaluation.py:301-306 + utils_flow/visualization_utils.py:75-77 on top of the two

torch itself is the oracle for these ops (oracle/dvd_oracle.py calls it); this file exists so that a `-m "not gpu"` test can
PIN the order the HIP kernels were written to against the torch build of the box it runs on
(tests/test_oracle_golden.py::test_aten_order_restatement): if a torch build ever contracts its CPU kernels differently,
that test fails before any GPU byte comparison is misread.  An FMA is emulated as ONE rounding of the exact float64
product-sum (24 + 24 bits of product fit a double exactly; the double rounding of the sum is not a concern at the sizes
tested: the comparison with torch is exact equality and would show it).  How the order was found: tests/tools/aten_order_probe.py."""
import numpy as np

f32 = np.float32


def _fma(a, b, c):
    return (np.asarray(a, f32).astype(np.float64) * np.asarray(b, f32).astype(np.float64)
            + np.asarray(c, f32).astype(np.float64)).astype(f32)


def _mul(a, b):
    return (np.asarray(a, f32) * np.asarray(b, f32)).astype(f32)


def _axis(n_in, n_out):
    """ATen compute_source_index_and_lambda, align_corners=True: (i0, i1, l0, l1) per output index"""
    scale = f32(n_in - 1) / f32(n_out - 1) if n_out > 1 else f32(0)
    src = (scale * np.arange(n_out, dtype=f32)).astype(f32)
    i0 = np.minimum(src.astype(np.int64), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    l1 = np.clip((src - i0.astype(f32)).astype(f32), f32(0), f32(1))
    return i0, i1, (f32(1) - l1).astype(f32), l1


def interpolate_ac(x, H, W):
    """x [..., h, w] float32 (NCHW-contiguous, at most 3 channels) -> [..., H, W].  torch picks one of TWO CPU kernels by the
    OUTPUT size (UpSampleKernel.cpp, _use_vectorized_kernel_cond_2d): H + W <= 128 takes the channels-last kernel - the four
    weights are products of the axis weights and the taps are summed by three FMAs; larger outputs take the generic separable one."""
    x = np.asarray(x, f32)
    y0, y1, ly0, ly1 = _axis(x.shape[-2], H)
    x0, x1, lx0, lx1 = _axis(x.shape[-1], W)
    a, b = x[..., y0, :][..., x0], x[..., y0, :][..., x1]
    c, d = x[..., y1, :][..., x0], x[..., y1, :][..., x1]
    if H + W <= 128:
        w00, w01 = _mul(ly0[:, None], lx0[None, :]), _mul(ly0[:, None], lx1[None, :])
        w10, w11 = _mul(ly1[:, None], lx0[None, :]), _mul(ly1[:, None], lx1[None, :])
        return _fma(d, w11, _fma(c, w10, _fma(a, w00, _mul(b, w01))))
    t0 = _fma(a, lx0, _mul(b, lx1))
    t1 = _fma(c, lx0, _mul(d, lx1))
    return _fma(t0, ly0[:, None], _mul(t1, ly1[:, None]))


def grid_sample_ac(src, grid_nchw):
    """src [n, c, hin, win], grid [n, 2, h, w] (channel 0 = x) -> [n, c, h, w]; finite coordinates"""
    src, grid = np.asarray(src, f32), np.asarray(grid_nchw, f32)
    n, c, hin, win = src.shape
    x = _mul(grid[:, 0] + f32(1), f32(win - 1) / f32(2))
    y = _mul(grid[:, 1] + f32(1), f32(hin - 1) / f32(2))
    xw, yn = np.floor(x), np.floor(y)
    xe, ys = (xw + f32(1)).astype(f32), (yn + f32(1)).astype(f32)
    w, e, nn, s = (x - xw).astype(f32), (xe - x).astype(f32), (y - yn).astype(f32), (ys - y).astype(f32)
    out = np.empty((n, c) + x.shape[1:], f32)
    for b in range(n):
        def tap(xi, yi):
            ok = (xi >= 0) & (xi <= win - 1) & (yi >= 0) & (yi <= hin - 1)
            v = src[b][:, np.clip(yi, 0, hin - 1).astype(np.int64), np.clip(xi, 0, win - 1).astype(np.int64)]
            return np.where(ok[None], v, f32(0))
        acc = _mul(tap(xw[b], yn[b]), _mul(s[b], e[b])[None])
        acc = _fma(tap(xe[b], yn[b]), _mul(s[b], w[b])[None], acc)
        acc = _fma(tap(xw[b], ys[b]), _mul(nn[b], e[b])[None], acc)
        out[b] = _fma(tap(xe[b], ys[b]), _mul(nn[b], w[b])[None], acc)
    return out


def unwarp_tail(flow, src_f32, scale=0.987):
    """flow [1, 2, G, G], src [1, 3, H, W] -> (grid [1, 2, H, W], out_f32 [H, W, 3], out_u8 [H, W, 3])"""
    H, W = src_f32.shape[-2:]
    s = interpolate_ac(flow, H, W)
    k = np.arange(512, dtype=f32) / f32(511)
    base512 = np.stack([np.broadcast_to(k[None, :], (512, 512)), np.broadcast_to(k[:, None], (512, 512))])[None]
    base = interpolate_ac(base512, H, W)
    t = (s + base).astype(f32)
    grid = _mul((_mul(_mul(t, f32(1)), f32(2)) - f32(1)).astype(f32), f32(scale))
    out = grid_sample_ac(src_f32, grid)[0].transpose(1, 2, 0)
    return grid, out, out.astype(np.uint8)
