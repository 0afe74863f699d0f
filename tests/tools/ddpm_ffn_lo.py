#!/usr/bin/env python3
"""(Checker-side experiment.)  250-step DDPM at G = 16 against the oracle with and without the decoder-FFN lo pass
(engine option ffn_lo): the number behind keeping ffn_lo = 1 as the default.  usage: python tests/tools/ddpm_ffn_lo.py"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch  # noqa: E402

from dvd_amd import sampler, schedule, synth  # noqa: E402
from dvd_amd.engine import Engine  # noqa: E402
from oracle import dvd_oracle as O  # noqa: E402

grid, steps = 16, 250
sd = synth.synth_state_dict(grid, 7, blocks=[11])
d0 = synth.synth_document(0, grid, 1234)
keys = ("y512", "mask_cat", "mask_y512", "line_msk")
doc = {k: torch.from_numpy(d0[k])[None] for k in keys}
xT = torch.from_numpy(synth.synth_noise(0, 2, grid, 1234))
noises = {i: torch.from_numpy(synth.synth_noise(0, 2, grid, 1234, step=i)) for i in range(steps)}
ref = O.Oracle(sd, grid).sample_loop(O.Schedule(steps), xT, doc, sampler="ddpm", noises=noises)
tab = schedule.Tables(schedule.named_betas("cosine", steps))
eng = Engine(grid, 1, 2)
eng.load_state_dict(sd)
eng.prepare(*(doc[k].cuda() for k in keys))
for name, opts in (("full split", {}), ("ffn_lo = 0", {"ffn_lo": 0}), ("split_weights = 0", {"split_weights": 0})):
    eng.set_option("ffn_lo", 1); eng.set_option("split_weights", 1)
    for k, v in opts.items():
        eng.set_option(k, v)
    out = sampler.sample(eng, tab, xT.cuda(), sampler="ddpm", noise_fn=lambda i: noises[i].cuda())
    print(f"250-step DDPM, G=16, {name:18s}: coordinate RMSE vs oracle {float((out.cpu() - ref).pow(2).mean().sqrt()):.3e}")
