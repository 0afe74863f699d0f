#!/usr/bin/env python3
"""The DiT block's fc1 GEMM at the bench's shape (M = 4 x 16 x 20736, N = 1536, K = 384, bias + GELU(tanh), f16 out)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from dvd_amd import ops
M, N, K = 4 * 16 * 20736, 1536, 384
a = torch.randn(M, K, device="cuda").half()
w = (torch.randn(N, K, device="cuda") * 0.05).half()
b = torch.randn(N, device="cuda") * 0.1
out = torch.empty(M, N, dtype=torch.float16, device="cuda")
for act in (1, 0):
    f = lambda: ops.gemm_nt(a, w, out16=out, bias=b, act=act)
    for _ in range(2): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for x, y in ev:
        x.record(); f(); y.record()
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in ev)[3]
    print(f"fc1 act={act}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TF/s")
ref = torch.nn.functional.gelu(a[:4096].float() @ w.float().t() + b, approximate="tanh")
ops.gemm_nt(a[:4096].contiguous(), w, out16=out[:4096], bias=b, act=1)
print("max abs err vs torch gelu(tanh) on 4096 rows:", float((out[:4096].float() - ref).abs().max()))
