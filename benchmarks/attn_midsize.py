#!/usr/bin/env python3
"""Where should the 256-row generated kernels take over from the 128-row glds kernels?  Decoder (6 x 256) and block (6 x 64)
attention at the token counts of the grids G = 64 .. 288 (T = (G/2)^2), 16 samples, lab build: forced generated kernel vs forced
glds kernel.  usage: python benchmarks/attn_midsize.py"""
import os, sys, statistics
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; _lab.use_lab()
import torch
from dvd_amd import ops
B = 16
for hd, force_gen, force_old in ((256, "DVD_ATTN_R64", "DVD_ATTN_R32"), (64, "DVD_ATTN_H64X", "DVD_ATTN_GLDS64")):
    for G in (64, 96, 128, 160, 192, 224, 288):
        T = (G // 2) ** 2
        if T % 64:
            T = T // 64 * 64
        C = 6 * hd
        qk = torch.randn(B, T, 2 * C, device="cuda").half(); vt = torch.randn(B, C, T, device="cuda").half()
        out = torch.empty(B, T, C, dtype=torch.float16, device="cuda")
        res = {}
        for name, env in (("generated", force_gen), ("glds", force_old)):
            for e in (force_gen, force_old):
                os.environ.pop(e, None)
            os.environ[env] = "1"
            f = lambda: ops.flash_attn(qk[:, :, :C], qk[:, :, C:], vt, out, 6, hd, 1.0 / hd ** 0.5)
            for _ in range(3): f()
            torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            res[name] = statistics.median(ts)
        fl = 4.0 * T * T * C * B
        print(f"hd {hd:3d}  G {G:3d}  T {T:5d}: generated {res['generated']:8.3f} ms = {fl / res['generated'] / 1e9:5.0f} TF/s   "
              f"glds {res['glds']:8.3f} ms = {fl / res['glds'] / 1e9:5.0f} TF/s   ratio {res['glds'] / res['generated']:.3f}")
        del qk, vt, out
