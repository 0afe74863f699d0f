#!/usr/bin/env python3
"""The DiT block's V^T projections at the bench shape (weights [384, 384] on the A side, N = T = 20736 tokens, K = 384, row
bias): self-attention (batch 4 x 16 = 64) and the r-stream's cross-attention (batch 16).  Short K, every token panel read
by two tiles only: latency-bound.  usage: python benchmarks/gemm_vt384_time.py [--lab]   (DVD_GEMM_RING=1 / small tiles)"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; LIBSEL = _lab.which()
import torch
from dvd_amd import ops
T, D = 20736, 384
w = (torch.randn(D, D, device="cuda") * 0.05).half()
bias = torch.randn(D, device="cuda") * 0.1
for B in (64, 16):
    h = torch.randn(B * T, D, device="cuda").half()
    vt = torch.empty(B, D, T, dtype=torch.float16, device="cuda")
    for small in (False, True):
        f = lambda: ops.gemm_nt(w, h, out16=vt.view(B * D, T), bias=bias, bias_row=True, batch=B, M=D, N=T, K=D, lda=D, ldb=D,
                                strides={"B": T * D, "C16": D * T}, small_tiles=small)
        for _ in range(2): f()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
        for a, b in ev:
            a.record(); f(); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)[3]
        print(f"V^T 384 batch {B} small_tiles={small}: {ms:.3f} ms  {2.0 * B * T * D * D / ms / 1e9:.0f} TF/s  lib={LIBSEL} ring={os.environ.get('DVD_GEMM_RING', '-')}")
    del h, vt
