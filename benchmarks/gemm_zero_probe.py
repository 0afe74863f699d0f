#!/usr/bin/env python3
"""Is the 256 x 256 GEMM held by power or by the fabric?  The same launch on random and on all-zero operands (zero operands
draw far less power: the chip holds a higher clock; a fabric-bound kernel does not get faster with it)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from dvd_amd import ops
M = 331776
for name, N, K in (("qk  N=3072 K=1536", 3072, 1536), ("fc  N=1536 K=1536", 1536, 1536)):
    for kind in ("random", "zeros", "random"):
        a = (torch.randn(M, K, device="cuda") if kind == "random" else torch.zeros(M, K, device="cuda")).half()
        w = (torch.randn(N, K, device="cuda") * 0.05 if kind == "random" else torch.zeros(N, K, device="cuda")).half()
        out = torch.empty(M, N, dtype=torch.float16, device="cuda")
        f = lambda: ops.gemm_nt(a, w, out16=out)
        for _ in range(3): f()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
        for x, y in ev:
            x.record(); f(); y.record()
        torch.cuda.synchronize()
        ms = sorted(x.elapsed_time(y) for x, y in ev)[4]
        print(f"{name} {kind:6s}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TF/s")
        del a, w, out
