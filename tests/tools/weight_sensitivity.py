#!/usr/bin/env python3
"""(Checker-side experiment.)  How much does the engine's coordinate error depend on the synthetic weight draw?  No trained
checkpoint is reachable offline, so the split-weight margin is probed on several weight families instead of one:
other seeds, larger weights (gain 1.5 on every matrix), and a heavy-tailed family (|u|^3-shaped: a few large weights,
many tiny ones - closer to a trained network's spectrum).  50-step DDIM at G = 72 (a large-tile grid: the weight dithering
is live) against the oracle, TAME output gain (x0 stays inside (-1, 1)), un-clamped last-step x0 RMSE with the dithered
weights (round 3's default), the (hi, lo) split in every GEMM (round 2's default) and plain f16.
usage: python tests/tools/weight_sensitivity.py > profiles/<round>_weight_sensitivity.txt"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dvd_amd import sampler, schedule, synth  # noqa: E402
from dvd_amd.engine import Engine  # noqa: E402
from oracle import dvd_oracle as O  # noqa: E402

grid, steps = 72, 50
torch.set_num_threads(min(32, os.cpu_count() or 8))
d0 = synth.synth_document(0, grid, 1234)
keys = ("y512", "mask_cat", "mask_y512", "line_msk")
doc = {k: torch.from_numpy(d0[k])[None] for k in keys}
xT = torch.from_numpy(synth.synth_noise(0, 1, grid, 1234))
tab = schedule.Tables(schedule.named_betas("cosine", steps))
spec = synth.state_dict_spec(grid, [11])


def family(name):
    gain = synth.tame_gain(steps)
    if name.startswith("seed"):
        return synth.synth_state_dict(grid, int(name[4:]), blocks=[11], out_gain=gain)
    sd = synth.synth_state_dict(grid, 7, blocks=[11], out_gain=gain)
    for k, (shape, kind) in spec.items():
        if kind not in ("w", "w_mod") or len(shape) < 2:
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
    tr_ref = []
    with torch.no_grad():
        O.Oracle(sd, grid).sample_loop(O.Schedule(steps), xT, doc, trace=tr_ref)
    eng = Engine(grid, 1, 1)
    eng.load_state_dict(sd)
    eng.prepare(*(doc[k].cuda() for k in keys))
    res = {}
    for tag, opts in (("dither", {"dither": 1, "split_weights": 1}), ("split", {"dither": 0, "split_weights": 1}),
                      ("plain f16", {"dither": 0, "split_weights": 0})):
        for k, v in opts.items():
            eng.set_option(k, v)
        tr = []
        sampler.sample(eng, tab, xT.cuda(), trace=tr)
        res[tag] = float((tr[-1].cpu() - tr_ref[-1]).pow(2).mean().sqrt())
    print(f"{name:11s}: un-clamped last-x0 RMSE  dither {res['dither']:.3e}   split {res['split']:.3e}   plain f16 "
          f"{res['plain f16']:.3e}   (last x0 std {float(tr_ref[-1].std()):.3f}, saturated "
          f"{float((tr_ref[-1].abs() >= 1).float().mean()):.4f})", flush=True)
    del eng
