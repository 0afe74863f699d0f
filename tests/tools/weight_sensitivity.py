#!/usr/bin/env python3
"""(Checker-side experiment.)  How much does the engine's coordinate error depend on the synthetic weight draw?  No trained
checkpoint is reachable offline, so the split-weight margin is probed on several weight families instead of one:
other seeds, larger weights (gain 1.5 on every matrix), and a heavy-tailed family (|u|^3-shaped: a few large weights,
many tiny ones - closer to a trained network's spectrum).  50-step DDIM at G = 32 against the oracle, with the full
split and with the lo parts off.  usage: python tests/tools/weight_sensitivity.py"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dvd_amd import sampler, schedule, synth  # noqa: E402
from dvd_amd.engine import Engine  # noqa: E402
from oracle import dvd_oracle as O  # noqa: E402

grid, steps = 32, 50
torch.set_num_threads(min(32, os.cpu_count() or 8))
d0 = synth.synth_document(0, grid, 1234)
keys = ("y512", "mask_cat", "mask_y512", "line_msk")
doc = {k: torch.from_numpy(d0[k])[None] for k in keys}
xT = torch.from_numpy(synth.synth_noise(0, 2, grid, 1234))
tab = schedule.Tables(schedule.named_betas("cosine", steps))
spec = synth.state_dict_spec(grid, [11])


def family(name):
    if name.startswith("seed"):
        return synth.synth_state_dict(grid, int(name[4:]), blocks=[11])
    sd = synth.synth_state_dict(grid, 7, blocks=[11])
    for k, (shape, kind) in spec.items():
        if kind not in ("w", "w_mod", "w_out") or len(shape) < 2:
            continue
        w = np.asarray(sd[k], dtype=np.float64)
        if name == "gain1.5":
            sd[k] = (w * 1.5).astype(np.float32)
        elif name == "heavy_tail":                      # same second moment, kurtosis ~ 9x uniform's
            a = np.abs(w).max() + 1e-30
            t = np.sign(w) * (np.abs(w) / a) ** 3
            sd[k] = (t * (np.sqrt((w ** 2).mean()) / (np.sqrt((t ** 2).mean()) + 1e-30))).astype(np.float32)
    return sd


for name in ("seed7", "seed8", "seed9", "gain1.5", "heavy_tail"):
    sd = family(name)
    ref = O.Oracle(sd, grid).sample_loop(O.Schedule(steps), xT, doc)
    eng = Engine(grid, 1, 2)
    eng.load_state_dict(sd)
    eng.prepare(*(doc[k].cuda() for k in keys))
    res = {}
    for tag, split in (("full split", 1), ("no split", 0)):
        eng.set_option("split_weights", split)
        out = sampler.sample(eng, tab, xT.cuda())
        res[tag] = float((out.cpu() - ref).pow(2).mean().sqrt())
    print(f"{name:11s}: coordinate RMSE full split {res['full split']:.3e}   un-split f16 {res['no split']:.3e}   "
          f"(map std {float(ref.std()):.3f})", flush=True)
    del eng
