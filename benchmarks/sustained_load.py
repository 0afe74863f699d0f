import os, sys, time, torch
sys.path.insert(0, ".")
from dvd_amd import ops
which = sys.argv[1]
if which == "gemm":
    M, N, K = 331776, 3072, 1536
    a = torch.randn(M, K, device="cuda").half(); w = torch.randn(N, K, device="cuda") * 0.05; hi = w.half(); lo = (w - hi.float()).half()
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    f = lambda: ops.gemm_nt(a, hi, out16=out, b_lo=lo, lo_scale=1.0)
else:
    B, T, C = 16, 20736, 1536
    qk = torch.randn(B, T, 2 * C, device="cuda").half(); vt = torch.randn(B, C, T, device="cuda").half()
    out = torch.empty(B, T, C, dtype=torch.float16, device="cuda")
    f = lambda: ops.flash_attn(qk[:, :, :C], qk[:, :, C:], vt, out, 6, 256, 1.0 / 16)
print("start", flush=True)
t0 = time.time()
while time.time() - t0 < float(sys.argv[2]):
    for _ in range(20): f()
    torch.cuda.synchronize()
print("done", flush=True)
