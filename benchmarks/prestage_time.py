#!/usr/bin/env python3
"""Per-document time of ingest + the pre-stage conditioning nets (evaluation.py:162-216) on the conv-net executor.
usage: python benchmarks/prestage_time.py [grid=64] [batch=1]   (batch documents per pass of the op lists)"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from dvd_amd import ops, prestage, synth
G = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tt = lambda sd: {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
dewarp, seg, line = prestage.GeoTr_Seg_Inf(), prestage.Seg(), prestage.UNet(n_channels=3, n_classes=1)
dewarp.msk.load_state_dict(tt(synth.synth_convnet_state_dict("u2netp", 11)), strict=True)
seg.load_state_dict(tt(synth.synth_convnet_state_dict("u2netp", 22, prefix="msk.")), strict=True)
line.load_state_dict(tt(synth.synth_convnet_state_dict("unet", 13)), strict=True)
for m in (dewarp, seg, line):
    m.to("cuda"); m.eval()
img = torch.from_numpy(np.ascontiguousarray((synth.smooth_image("pt/img", 1024, 768, 1).transpose(1, 2, 0) * 255).astype(np.uint8))).cuda()
def one():
    y = torch.stack([ops.ingest_u8(img, swap_rb=False, out_size=512) for _ in range(B)])
    return prestage.conditioning(dewarp, seg, line, y, G)
for _ in range(3): one()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); one(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
ev[0].record(); one(); ev[1].record(); torch.cuda.synchronize()
print(f"ingest + pre-stage nets, G={G}, {B} documents per pass: wall {sorted(ts)[5] / B:.2f} ms per document "
      f"(device busy span {ev[0].elapsed_time(ev[1]) / B:.2f} ms per document)")
