#!/usr/bin/env python3
"""Where a wave of gemm_nt_ring128_kernel spends its lifetime (lab build; s_memtime runs at 100 MHz: 10 ns per tick):
prologue (start -> first two slabs landed), K loop, epilogue.  usage: python benchmarks/gemm_ring128_stamps.py [M N K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; _lab.use_lab()
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import ctypes as C
import torch
from dvd_amd import lib, ops
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2048, 1536, 1536)
a = torch.randn(M, K, device="cuda").half()
w = torch.randn(N, K, device="cuda") * 0.05
hi = w.half(); lo = ((w - hi.float()) * 2048.0).half()
out = torch.empty(M, N, dtype=torch.float32, device="cuda")
nblk = ((M + 127) // 128) * ((N + 127) // 128)
st = torch.zeros(nblk * 4 * 8, dtype=torch.int64, device="cuda")
lib.call("dvd_gemm_debug_stamps", C.c_void_p(st.data_ptr()))
for split in (True, False):
    for _ in range(3):
        ops.gemm_nt(a, hi, b_lo=lo if split else None, out32=out, small_tiles=True)
    torch.cuda.synchronize()
    s = st.view(nblk, 4, 8).cpu().double()
    pro, loop, epi = s[:, :, 1] - s[:, :, 0], s[:, :, 2] - s[:, :, 1], s[:, :, 3] - s[:, :, 2]
    slabs = K // 64 * (2 if split else 1)
    span = (s[:, :, 3].max() - s[:, :, 0].min())
    print(f"{M}x{N}x{K} {'(hi, lo)' if split else 'single'}: per wave, ticks of s_memtime: prologue {pro.mean():.0f}  K loop {loop.mean():.0f} = {loop.mean() / slabs:.2f} per slab "
          f"({slabs} slabs)  epilogue {epi.mean():.0f}   first start -> last end {span:.0f}")
lib.call("dvd_gemm_debug_stamps", C.c_void_p(0))
