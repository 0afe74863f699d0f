#!/usr/bin/env python3
"""DVD_ATTN_DEBUG=1 python benchmarks/attn_stamps.py : per-phase cycle shares of a KV tile (head_dim 256) of the ROUND 1-3
kernels (r64 of rounds 1-3 by default - now DVD_ATTN_R64OLD -, DVD_ATTN_R32 / DVD_ATTN_PIPE / DVD_ATTN_BULK for the older
ones).  The production kernel's stamps: benchmarks/attn_stamps_r64p.py."""
import os, sys
if not (os.environ.get("DVD_ATTN_R32") or os.environ.get("DVD_ATTN_PIPE") or os.environ.get("DVD_ATTN_BULK")):
    os.environ["DVD_ATTN_R64OLD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; _lab.use_lab()   # the DVD_* switches exist in the lab build only (make -C dvd_amd/csrc lab)
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import ctypes as C
import torch
from dvd_amd import lib, ops
hd = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B, T = (2 if hd == 256 else 8), 20736
Cc = 6 * hd
qk = torch.randn(B, T, 2 * Cc, device="cuda").half(); vt = torch.randn(B, Cc, T, device="cuda").half()
out = torch.empty(B, T, Cc, dtype=torch.float16, device="cuda")
nwg = (T // 128) * 6 * B
if not os.environ.get("DVD_ATTN_R32") and not os.environ.get("DVD_ATTN_PIPE") and not os.environ.get("DVD_ATTN_BULK"):
    nwg = (T // 256) * 6 * B
st = torch.zeros(nwg * 4 * 5, dtype=torch.int64, device="cuda")
lib.call("dvd_attn_debug_stamps", C.c_void_p(st.data_ptr()))
for _ in range(2):
    ops.flash_attn(qk[:, :, :Cc], qk[:, :, Cc:], vt, out, 6, hd, 1.0 / (hd ** 0.5))
torch.cuda.synchronize()
per_tile = 64 if (os.environ.get("DVD_ATTN_R32") or os.environ.get("DVD_ATTN_PIPE") or os.environ.get("DVD_ATTN_BULK")) else 32
s = st.view(nwg * 4, 5).cpu().double() / (T // per_tile)
names = ["issue LDS-DMA (16 loads)", "S^T phase (32 MFMA)", "max/rescale + P chunk 0", "PV phase (32 MFMA) + softmax", "vmcnt(0) + barrier"]
if os.environ.get("DVD_ATTN_PIPE"):   # software-pipelined kernel: per 64-key tile = 2 blocks
    names = ["2 x (K prefetch issue + max tree + rescale test)", "2 x (16 S^T MFMA || 13 exps)", "2 x (16 PV MFMA || 3 exps, sa+sb)",
             "2 x vmcnt wait", "2 x s_barrier"]
tot = s.sum(1).mean()
for k, n in enumerate(names):
    print(f"{n:34s} {s[:, k].mean():8.0f} cycles/tile  {100 * s[:, k].mean() / tot:5.1f}%")
print(f"total {tot:.0f} cycles per tile (MFMA minimum {2048 if hd == 256 else 512})")
