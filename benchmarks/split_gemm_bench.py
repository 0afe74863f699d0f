#!/usr/bin/env python3
"""One-pass vs two-pass split-weight GEMM at the engine's decoder shapes (M = 16 x 20736)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; _lab.use_lab()   # the DVD_* switches exist in the lab build only (make -C dvd_amd/csrc lab)
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from dvd_amd import ops
M, K = 331776, 1536
a = torch.randn(M, K, device="cuda").half()
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
for N in (1536, 2048, 3072):
    w = torch.randn(N, K, device="cuda") * 0.05; hi = w.half(); lo = (w - hi.float()).half()
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    f = lambda: ops.gemm_nt(a, hi, out16=out, b_lo=lo, lo_scale=1.0)
    ms = t(f); os.environ["DVD_GEMM_TWOPASS"] = "1"; ms2 = t(f); os.environ.pop("DVD_GEMM_TWOPASS")
    fl = 2 * M * N * K * 2 / 1e12
    print(f"N={N}: one-pass {ms:.3f} ms ({fl / ms * 1e3:.0f} TF/s executed)   two-pass {ms2:.3f} ms ({fl / ms2 * 1e3:.0f} TF/s)")
