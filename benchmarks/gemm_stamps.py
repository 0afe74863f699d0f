#!/usr/bin/env python3
"""DVD_GEMM_DEBUG=3 python benchmarks/gemm_stamps.py : where a wave of the large-tile GEMM spends its lifetime."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; _lab.use_lab()   # the DVD_* switches exist in the lab build only (make -C dvd_amd/csrc lab)
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import ctypes as C
import torch
from dvd_amd import lib, ops
M, N, K = 331776, 1536, 1536
a = torch.randn(M, K, device="cuda").half(); b = torch.randn(N, K, device="cuda").half()
out = torch.empty(M, N, dtype=torch.float16, device="cuda")
nblk = (M // 256) * (N // 256)
st = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device="cuda")
lib.call("dvd_gemm_debug_stamps", C.c_void_p(st.data_ptr()))
for _ in range(2):
    ops.gemm_nt(a, b, out16=out)
torch.cuda.synchronize()
s = st.view(nblk, 8, 8).cpu().double()
pro, loop, epi, drain = s[:, :, 1] - s[:, :, 0], s[:, :, 2] - s[:, :, 1], s[:, :, 3] - s[:, :, 2], s[:, :, 4] - s[:, :, 3]
print("epilogue issue %.0f   store drain %.0f" % (epi.mean(), drain.mean()))
print("epilogue pieces: LDS writes A %.0f | reads+stores A %.0f | LDS writes B %.0f | reads+stores B %.0f" % ((s[:, :, 5] - s[:, :, 2]).mean(), (s[:, :, 6] - s[:, :, 5]).mean(), (s[:, :, 7] - s[:, :, 6]).mean(), (s[:, :, 3] - s[:, :, 7]).mean()))
print("ticks (100 MHz s_memtime? raw units) per wave: prologue %.0f  K-loop %.0f (per K-step %.1f)  epilogue %.0f" % (pro.mean(), loop.mean(), loop.mean() / (K // 64), epi.mean()))
tot = (s[:, :, 4] - s[:, :, 0]).mean()
print("shares: prologue %.1f%%  loop %.1f%%  epilogue %.1f%%" % (100 * pro.mean() / tot, 100 * loop.mean() / tot, 100 * epi.mean() / tot))
span = s[:, :, 3].max() - s[:, :, 0].min()
print("kernel span %.0f ticks; sum of wave lifetimes / (span * 8 waves * 256 CUs) = %.2f" % (span, (s[:, :, 3] - s[:, :, 0]).sum() / (span * 8 * 256)))
