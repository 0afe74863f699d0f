#!/usr/bin/env python3
"""A/B of two BUILDS of the library on one box: each build in its own child process, alternating, same random inputs
(seeded); prints the median attention time per build and round.  usage: attn_lib_ab.py <hd> <libA.so> <libB.so> [rounds=3]"""
import os, subprocess, sys, statistics
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys, statistics
sys.path.insert(0, os.path.abspath(os.path.join(sys.argv[3], "..")))
import torch
from dvd_amd import lib, ops
lib.use_library(sys.argv[1])
hd = int(sys.argv[2]); B, T = 16, 20736; C = 6 * hd
torch.manual_seed(0)
qk = torch.randn(B, T, 2 * C, device="cuda").half(); vt = torch.randn(B, C, T, device="cuda").half()
out = torch.empty(B, T, C, dtype=torch.float16, device="cuda")
f = lambda: ops.flash_attn(qk[:, :, :C], qk[:, :, C:], vt, out, 6, hd, 1.0 / hd ** 0.5)
for _ in range(3): f()
torch.cuda.synchronize()
ts = []
for _ in range(9):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(f"{statistics.median(ts):.3f} {float(out.float().abs().sum()):.6e}")
'''
hd, la, lb = sys.argv[1], os.path.abspath(sys.argv[2]), os.path.abspath(sys.argv[3])
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
res = {la: [], lb: []}
for r in range(rounds):
    for l in (la, lb):
        o = subprocess.run([sys.executable, "-c", CHILD, l, hd, HERE], capture_output=True, text=True)
        if o.returncode:
            print(o.stderr[-2000:]); sys.exit(1)
        ms, chk = o.stdout.split()[-2:]
        res[l].append(float(ms))
        print(f"round {r} {os.path.basename(l):28s} {ms} ms   checksum {chk}", flush=True)
fl = 4 * 20736 ** 2 * 6 * int(hd) * 16
for l in (la, lb):
    m = statistics.median(res[l])
    print(f"{os.path.basename(l):28s} median {m:.3f} ms = {fl / m / 1e9:.0f} TF/s")
