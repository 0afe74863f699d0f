#!/usr/bin/env python3
"""Lab A/B (DVD_GEMM_SPREAD): the two-sweep 256 x 256 kernel at 65536 rows with its LDS-DMA pieces as a burst (product) or spread
over the slab.  usage: python benchmarks/lab/big_spread_ab.py   (result: profiles/r5_gemm_big_spread.txt)"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "benchmarks")); import _lab; _lab.use_lab()
import torch
from dvd_amd import ops
M = 65536
def timed(f, reps=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for x, y in ev:
        x.record(); f(); y.record()
    torch.cuda.synchronize()
    return sorted(x.elapsed_time(y) for x, y in ev)[len(ev) // 2] * 1e3
for name, N, K in (("qk", 3072, 1536), ("c1", 2048, 1536), ("fc", 1536, 1536), ("c2", 1536, 2048)):
    a = torch.randn(M, K, device="cuda").half(); w = torch.randn(N, K, device="cuda") * 0.05
    hi = w.half(); lo = ((w - hi.float()) * 2048.0).half()
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    f = lambda: ops.gemm_nt(a, hi, b_lo=lo, out16=out, small_tiles=2)
    res = {}
    for tag in ("burst", "spread", "burst", "spread"):
        if tag == "spread": os.environ["DVD_GEMM_SPREAD"] = "1"
        else: os.environ.pop("DVD_GEMM_SPREAD", None)
        res.setdefault(tag, []).append(timed(f))
    os.environ.pop("DVD_GEMM_SPREAD", None)
    print(f"{name} rows {M}: two-sweep 256x256 kernel: burst {min(res['burst']):.1f} us  spread {min(res['spread']):.1f} us  ({4.0*M*N*K/min(res['burst'])/1e6:.0f} TF/s executed)")
