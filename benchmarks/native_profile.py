#!/usr/bin/env python3
"""The reference's operating point, 32 documents per engine batch (bench.py's native_point_batch32 leg), alone - for
`rocprofv3 --kernel-trace --stats -- python3 benchmarks/native_profile.py [docs=32] [reps=5]`: where the 9 ms per document go
(VERDICT r4 next-6).  Prints ms per batch / per document."""
import os, statistics, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; LIBSEL = _lab.which()   # --lab selects the lab build
import numpy as np
import torch
from dvd_amd import ops, prestage, sampler, schedule, synth
from dvd_amd.engine import Engine

Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
Gn, Sn, H = 64, 3, 2
sd = synth.synth_state_dict(Gn, seed=7, blocks=[11])
tt = lambda d: {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}  # noqa: E731
dewarp, seg, line = prestage.GeoTr_Seg_Inf(), prestage.Seg(), prestage.UNet(n_channels=3, n_classes=1)
dewarp.msk.load_state_dict(tt(synth.synth_convnet_state_dict("u2netp", 11)), strict=True)
seg.load_state_dict(tt(synth.synth_convnet_state_dict("u2netp", 22, prefix="msk.")), strict=True)
line.load_state_dict(tt(synth.synth_convnet_state_dict("unet", 13)), strict=True)
for m in (dewarp, seg, line):
    m.to(dev)
    m.eval()
img = synth.smooth_image("bench/native", 1024, 768, seed=1234)
img_u8 = torch.from_numpy(np.ascontiguousarray((img.transpose(1, 2, 0) * 255.0).astype(np.uint8))).to(dev)
imgs = [torch.roll(img_u8, shifts=(17 * d, 29 * d), dims=(0, 1)).contiguous() for d in range(Bn)]
eng = Engine(Gn, Bn, H, device=dev)
eng.load_state_dict(sd)
xT = torch.cat([torch.from_numpy(synth.synth_noise(d, H, Gn, 1234)).to(dev) for d in range(Bn)])
tab = schedule.Tables(schedule.named_betas("cosine", Sn))
stage = {}


def run():
    t = [time.perf_counter()]
    def mark():
        torch.cuda.synchronize(); t.append(time.perf_counter())
    ys, srcs = zip(*[ops.ingest_u8(im, swap_rb=False, out_size=512, want_rgb=True) for im in imgs]); y = torch.stack(ys); mark()
    c = prestage.conditioning(dewarp, seg, line, y, Gn); mark()
    eng.prepare(y, c["mask_cat"].contiguous(), c["mask_y512"].contiguous(), c["line_msk"].contiguous()); mark()
    fl = sampler.sample(eng, tab, xT); mark()
    o8 = ops.unwarp_u8_batch(fl, torch.stack(srcs)); mark()
    for name, a, b in zip(("ingest", "prestage nets", "prepare_docs", "sampling", "unwarp"), t, t[1:]):
        stage.setdefault(name, []).append((b - a) * 1e3)
    return t[-1] - t[0]


for _ in range(2):
    run()
stage.clear()
lat = [run() * 1e3 for _ in range(reps)]
med = statistics.median(lat)
print(f"{Bn} documents per batch: {med:.2f} ms per batch = {med / Bn:.3f} ms per document = {Bn * 1e3 / med:.1f} documents/s "
      f"(stage-synchronised: adds a few launch gaps)")
for k, v in stage.items():
    print(f"  {k:14s} {statistics.median(v):8.2f} ms per batch  {statistics.median(v) / Bn:7.3f} ms per document")
