#!/usr/bin/env python3
"""DVD_GEMM_T384_DBG=5 python benchmarks/gemm_t384_stamps.py [N K] : where a wave of gemm_nt_t384_kernel spends its lifetime
(s_memtime: prologue = tile start -> first half slab landed, K loop, epilogue issue, store drain)."""
import os, sys
os.environ.setdefault("DVD_GEMM_T384_DBG", "5")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; _lab.use_lab()
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import ctypes as C
import torch
from dvd_amd import lib, ops
M = 331776
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1536, 1536)
MODE = sys.argv[3] if len(sys.argv) > 3 else "f16"          # f16 | f32 | res (f32 output + residual, in place)
a = torch.randn(M, K, device="cuda").half(); b = torch.randn(N, K, device="cuda").half()
out = torch.empty(M, N, dtype=torch.float16 if MODE == "f16" else torch.float32, device="cuda")
nblk = (M // 384) * (N // 256)
st = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device="cuda")
lib.call("dvd_gemm_debug_stamps", C.c_void_p(st.data_ptr()))
for _ in range(2):
    if MODE == "f16": ops.gemm_nt(a, b, out16=out)
    elif MODE == "f32": ops.gemm_nt(a, b, out32=out)
    else: ops.gemm_nt(a, b, out32=out, res=out)
torch.cuda.synchronize()
s = st.view(nblk, 8, 8).cpu().double()
pro, loop, epi, drain = s[:, :, 1] - s[:, :, 0], s[:, :, 2] - s[:, :, 1], s[:, :, 3] - s[:, :, 2], s[:, :, 4] - s[:, :, 3]
nh = K // 32
print(f"N={N} K={K} {MODE}: per wave and tile: prologue {pro.mean():.0f}  K loop {loop.mean():.0f} = {loop.mean() / nh:.1f} per half slab (MFMA minimum 1536)  "
      f"epilogue {epi.mean():.0f}  store drain {drain.mean():.0f}")
tot = (s[:, :, 4] - s[:, :, 0]).mean()
print("shares: prologue %.1f%%  loop %.1f%%  epilogue %.1f%%  drain %.1f%%" % (100 * pro.mean() / tot, 100 * loop.mean() / tot, 100 * epi.mean() / tot, 100 * drain.mean() / tot))
