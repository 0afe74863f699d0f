#!/usr/bin/env python3
"""Is the small-row GEMM bound by fabric round trips?  The same launch (gemm_nt_ring128_kernel: 2048 x N x 1536, weight pair) timed
(a) back to back - operands of a small N stay in the XCDs' L2s - and (b) after a 1 GiB fill that evicts L2 and the Infinity Cache.
usage: python benchmarks/gemm_l2warm.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; LIBSEL = _lab.which()
import torch
from dvd_amd import ops
M, K = 2048, 1536
junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda")          # 1 GiB
for N in (256, 512, 1536, 3072):
    a = torch.randn(M, K, device="cuda").half()
    w = torch.randn(N, K, device="cuda") * 0.05
    hi = w.half(); lo = ((w - hi.float()) * 2048.0).half()
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    f = lambda: ops.gemm_nt(a, hi, b_lo=lo, out32=out, small_tiles=True)
    for _ in range(3): f()
    def t(evict):
        ts = []
        for _ in range(12):
            if evict: junk.fill_(1.0)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); f(); e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        return sorted(ts)[len(ts) // 2]
    warm, cold = t(False), t(True)
    tiles = (M // 128) * (N // 128)
    print(f"N {N:5d} ({tiles:3d} tiles, weight pair {N * K * 4 / 1e6:5.1f} MB): back to back {warm:6.1f} us ({4.0 * M * N * K / warm / 1e6:5.0f} TF/s executed)   "
          f"after eviction {cold:6.1f} us   per 64-deep slab {warm / (2 * K / 64) * 1e3:5.0f} / {cold / (2 * K / 64) * 1e3:5.0f} ns   lib={LIBSEL}")
