#!/usr/bin/env python3
"""Soak of the generated attention kernels (lab build): spikes that force the deferred rescale in every tile variant (t % 6),
in the first and the last tiles, several at once, on both row/query blocks; ragged query counts; shared K/V.
Compared with float64 softmax(QK^T)V.  usage: attn_soak.py [ENVVAR ...]   (e.g. DVD_ATTN_R64, DVD_ATTN_R64M, DVD_ATTN_H64M)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; _lab.use_lab()
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from dvd_amd import ops
for e in sys.argv[1:]:
    os.environ[e] = "1"
hd = 64 if "DVD_ATTN_H64M" in sys.argv[1:] else 256
heads = 2
C = heads * hd
gen = torch.Generator(device="cpu").manual_seed(1234)
worst = 0.0
n = 0
for case in range(40):
    B = 1 + case % 3
    Bkv = B if case % 4 else 1
    if B % Bkv: Bkv = B
    tq = [64, 200, 256, 300, 513][case % 5]
    tk = 64 * (1 + (case * 7) % 13)
    q = torch.randn(B, tq, C, generator=gen).half()
    k = torch.randn(Bkv, tk, C, generator=gen).half()
    v = torch.randn(Bkv, tk, C, generator=gen).half()
    # spikes: keys that dominate chosen query rows by a lot (forces the rescale at the spike's tile)
    nsp = case % 4
    for sp in range(nsp):
        key = int(torch.randint(0, tk, (1,), generator=gen))
        row = int(torch.randint(0, tq, (1,), generator=gen))
        hh = int(torch.randint(0, heads, (1,), generator=gen))
        k[0, key, hh * hd:(hh + 1) * hd] = q[0, row, hh * hd:(hh + 1) * hd] * (3 + 2 * sp)
    scale = 1.0 / hd ** 0.5
    out = torch.zeros(B, tq, C, dtype=torch.float16, device="cuda")
    ops.flash_attn(q.cuda(), k.cuda(), v.transpose(1, 2).contiguous().cuda(), out, heads, hd, scale, kv_batch_div=B // Bkv)
    qh = q.double().reshape(B, tq, heads, hd).transpose(1, 2)
    kh = k.double().reshape(Bkv, tk, heads, hd).transpose(1, 2).repeat_interleave(B // Bkv, 0)
    vh = v.double().reshape(Bkv, tk, heads, hd).transpose(1, 2).repeat_interleave(B // Bkv, 0)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) @ vh).transpose(1, 2).reshape(B, tq, C)
    err = (out.cpu().double() - ref).abs().max().item()
    mag = ref.abs().max().item()
    worst = max(worst, err / max(1.0, mag))
    n += 1
    if err > 3e-3 * max(1.0, mag) or not torch.isfinite(out).all():
        print(f"FAIL case {case}: B={B} Bkv={Bkv} tq={tq} tk={tk} spikes={nsp}: err {err:.3e} (ref max {mag:.2f})")
print(f"{' '.join(sys.argv[1:]) or 'product'}: {n} cases, worst relative error {worst:.3e}")
