#!/usr/bin/env python3
"""A/B the head_dim-256 decoder attention kernels of the LAB build in ONE process, interleaved rounds on random data
(guide rules 24/25): the lab library reads its DVD_ATTN_* switches at every call, so the variants alternate launch by launch.
usage: python benchmarks/attn_ab.py [B=16] [rounds=7] [hd=256] [variant=ENV[,ENV...] ...]
       e.g.  python benchmarks/attn_ab.py 16 7 256 r64p= r64old=DVD_ATTN_R64OLD"""
import os, sys, statistics
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab
if "--lib" in sys.argv:                      # another build of the library (benchmarks/lab/alt/: generator experiments)
    _lab.which()
else:
    _lab.use_lab()
import torch
from dvd_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
hd = int(sys.argv[3]) if len(sys.argv) > 3 else 256
specs = sys.argv[4:] or ["r64x=", "r64m=DVD_ATTN_R64M"]
variants = []
for sp in specs:
    name, _, envs = sp.partition("=")
    variants.append((name, [e for e in envs.split(",") if e]))
ALL = sorted({e.split(":")[0] for _, es in variants for e in es})          # ENVVAR or ENVVAR:value
T, C = 20736, 6 * hd
qk = torch.randn(B, T, 2 * C, device="cuda").half()
vt = torch.randn(B, C, T, device="cuda").half()
outs = {n: torch.empty(B, T, C, dtype=torch.float16, device="cuda") for n, _ in variants}


def call(name, envs):
    for e in ALL:
        os.environ.pop(e, None)
    for e in envs:
        k, _, v = e.partition(":")
        os.environ[k] = v or "1"
    ops.flash_attn(qk[:, :, :C], qk[:, :, C:], vt, outs[name], 6, hd, 1.0 / hd ** 0.5)


for n, es in variants:
    call(n, es)
torch.cuda.synchronize()
ref = outs[variants[0][0]].float()
for n, _ in variants[1:]:
    d = (outs[n].float() - ref).abs().max().item()
    print(f"max |{n} - {variants[0][0]}| = {d:.3e}   (ref abs max {ref.abs().max().item():.3f})")
# warm the clocks: 2 s of back-to-back launches
for _ in range(20):
    call(*variants[0])
torch.cuda.synchronize()
ms = {n: [] for n, _ in variants}
for r in range(rounds):
    for n, es in variants:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            call(n, es)
        b.record()
        torch.cuda.synchronize()
        ms[n].append(a.elapsed_time(b) / 3)
fl = 4.0 * T * T * C * B
for n, _ in variants:
    med, mn = statistics.median(ms[n]), min(ms[n])
    print(f"{n:>10}: median {med:8.3f} ms = {fl / med / 1e9:6.0f} TF/s   min {mn:8.3f} ms = {fl / mn / 1e9:6.0f} TF/s   all {[round(x, 2) for x in ms[n]]}")
