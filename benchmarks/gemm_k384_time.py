#!/usr/bin/env python3
"""The DiT block's K = 384 GEMMs at the bench shape (M = 4 x 16 x 20736 rows): self-attention Q|K projection (N = 768, bias) and
fc1 (N = 1536, bias + GELU), 256x256 kernel vs 128x128 kernel (small_tiles)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from dvd_amd import ops
M, K = 4 * 16 * 20736, 384
a = torch.randn(M, K, device="cuda").half()
for name, N, act in (("sa_wqk", 768, 0), ("fc1", 1536, 1)):
    w = (torch.randn(N, K, device="cuda") * 0.05).half()
    b = torch.randn(N, device="cuda") * 0.1
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    for small in (False, True):
        f = lambda: ops.gemm_nt(a, w, out16=out, bias=b, act=act, small_tiles=small)
        for _ in range(2): f()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
        for x, y in ev:
            x.record(); f(); y.record()
        torch.cuda.synchronize()
        ms = sorted(x.elapsed_time(y) for x, y in ev)[3]
        print(f"{name} N={N} small_tiles={small}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TF/s")
    del w, out
