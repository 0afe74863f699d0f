#!/usr/bin/env python3
"""Time the V^T projection of a decoder layer at the bench shape: per sample V^T[1536, T] = W_v[1536, 1536] . h^T, weights
(hi + lo) on the A side, batch 16.  usage: python benchmarks/gemm_vt_time.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; LIBSEL = _lab.which()   # --lab selects the lab build
import torch
from dvd_amd import ops
N, T, D = 16, 20736, 1536
w = torch.randn(D, D, device="cuda") * 0.05
hi = w.half(); lo = (w - hi.float()).half()
h = torch.randn(N * T, D, device="cuda").half()
vt = torch.empty(N, D, T, dtype=torch.float16, device="cuda")
f = lambda: ops.gemm_nt(hi, h, out16=vt.view(N * D, T), a_lo=lo, lo_scale=1.0, batch=N, M=D, N=T, K=D, lda=D, ldb=D,
                        strides={"B": T * D, "C16": D * T})
for _ in range(2): f()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
for a, b in ev:
    a.record(); f(); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)[3]
print(f"V^T projection (two-pass A_lo): {ms:.3f} ms  executed {4.0 * N * T * D * D / ms / 1e9:.0f} TF/s  lib={LIBSEL}")
