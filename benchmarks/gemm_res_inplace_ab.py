#!/usr/bin/env python3
"""Round 6: does the residual flavour of gemm_nt_t384_kernel care whether the f32 stream is updated IN PLACE (the engine's use:
out32 = res = z) or read from one buffer and written to another?   usage: python benchmarks/gemm_res_inplace_ab.py [reps=7] [--lab]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; LIBSEL = _lab.which()
import torch
from dvd_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
M = 331776
for name, N, K in (("fc  N=1536 K=1536", 1536, 1536), ("c2  N=1536 K=2048", 1536, 2048)):
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") * 0.02).half()
    z = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    z2 = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    variants = {"in place (out = res)": lambda: ops.gemm_nt(a, w, out32=z, res=z),
                "out of place (out != res)": lambda: ops.gemm_nt(a, w, out32=z2, res=z),
                "f32 output, no residual": lambda: ops.gemm_nt(a, w, out32=z2)}
    for rnd in range(2):
        for vn, f in variants.items():
            for _ in range(2): f()
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for x, y in ev:
                x.record(); f(); y.record()
            torch.cuda.synchronize()
            ms = sorted(x.elapsed_time(y) for x, y in ev)[len(ev) // 2]
            print(f"{name}  {vn:28s}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TF/s   lib={LIBSEL}")
