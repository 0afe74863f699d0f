#!/usr/bin/env python3
"""Per-phase s_memtime shares of one 32-key tile of flash_attn_r64p_kernel<1> (lab build, DVD_ATTN_DEBUG).
Diagnostic build: its stamps drain the LDS reads in flight, read its SHARES, not its length.  usage: [B=16]"""
import os, sys
os.environ["DVD_ATTN_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; _lab.use_lab()
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import ctypes as C
import torch
from dvd_amd import lib, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
hd, T = 256, 20736
Cc = 6 * hd
qk = torch.randn(B, T, 2 * Cc, device="cuda").half(); vt = torch.randn(B, Cc, T, device="cuda").half()
out = torch.empty(B, T, Cc, dtype=torch.float16, device="cuda")
nwg = (T // 256) * 6 * B
st = torch.zeros(nwg * 4 * 6, dtype=torch.int64, device="cuda")
lib.call("dvd_attn_debug_stamps", C.c_void_p(st.data_ptr()))
for _ in range(3):
    ops.flash_attn(qk[:, :, :Cc], qk[:, :, Cc:], vt, out, 6, hd, 1.0 / (hd ** 0.5))
torch.cuda.synchronize()
s = st.view(nwg * 4, 6).cpu().double() / (T // 32)
names = ["phase 1: 32 S^T MFMA || 24 exp units, 4 K pieces (+ 4 reads drained)", "phase 2a: 16 PV MFMA || 8 exp units (+ drain)",
         "s_waitcnt vmcnt(0)", "s_barrier", "phase 2b: 16 PV MFMA || max, test, l; 4 V pieces (+ drain)", "rare rescale branch"]
tot = s.sum(1).mean()
for k, n in enumerate(names):
    print(f"{n:60s} {s[:, k].mean():8.0f} cycles/tile  {100 * s[:, k].mean() / tot:5.1f}%   (min wave {s[:, k].min():.0f}, max wave {s[:, k].max():.0f})")
print(f"total {tot:.0f} cycles per tile (MFMA minimum 2048)")
