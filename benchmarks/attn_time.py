#!/usr/bin/env python3
"""Time dvd_flash_attn at the bench's launch shapes (T = 20736): TF/s for head_dim 64 and 256.
usage: python benchmarks/attn_time.py [hd=64] [B=16] [reps=5]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; LIBSEL = _lab.which()   # --lab selects the lab build
import torch
from dvd_amd import ops
hd = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
T, C = 20736, 6 * hd
qk = torch.randn(B, T, 2 * C, device="cuda").half()
vt = torch.randn(B, C, T, device="cuda").half()
out = torch.empty(B, T, C, dtype=torch.float16, device="cuda")
f = lambda: ops.flash_attn(qk[:, :, :C], qk[:, :, C:], vt, out, 6, hd, 1.0 / hd ** 0.5)
for _ in range(2): f()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in ev:
    a.record(); f(); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
fl = 4.0 * T * T * C * B
print(f"hd={hd} B={B} lib={LIBSEL}: median {ms[len(ms)//2]:.3f} ms  {fl / ms[len(ms)//2] / 1e9:.0f} TF/s  (min {ms[0]:.3f})")
