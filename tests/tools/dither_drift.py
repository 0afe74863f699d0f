#!/usr/bin/env python3
"""(Checker-side experiment: lives under tests/ because it uses the oracle.)  How the three weight treatments of the
256-wide GEMMs drift over a 50-step DDIM roll-out - dithered single f16 (round 3's default on large grids), (hi, lo)
split (round 2's default), plain f16 - on both synthetic weight families:
  part A  G = 96, 1 document x 1 hypothesis: each against the CPU oracle (un-clamped x0 RMSE per step);
  part B  G = 288 (BASELINE's grid): GPU only, dithered and plain f16 against the SPLIT engine (whose own distance to
          the oracle over the whole G = 288 loop is profiles/archive/r2_parity_g288.json: 6.2e-4 un-clamped, plain family).
usage: python tests/tools/dither_drift.py > profiles/<round>_dither_drift.json"""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch  # noqa: E402

from dvd_amd import sampler, schedule, synth  # noqa: E402
from dvd_amd.engine import Engine  # noqa: E402
from oracle import dvd_oracle as O  # noqa: E402

S = 50
torch.set_num_threads(min(32, os.cpu_count() or 8))
keys = ("y512", "mask_cat", "mask_y512", "line_msk")
MODES = (("dither", {"dither": 1, "split_weights": 1}), ("split", {"dither": 0, "split_weights": 1}),
         ("plain_f16", {"dither": 0, "split_weights": 0}))


def rollouts(grid, gain):
    sd = synth.synth_state_dict(grid, 7, blocks=[11], out_gain=gain)
    d0 = synth.synth_document(0, grid, 1234)
    doc = {k: torch.from_numpy(d0[k])[None] for k in keys}
    xT = torch.from_numpy(synth.synth_noise(0, 1, grid, 1234))
    eng = Engine(grid, 1, 1)
    eng.load_state_dict(sd)
    eng.prepare(*(doc[k].cuda() for k in keys))
    tab = schedule.Tables(schedule.named_betas("cosine", S))
    traces = {}
    for name, opts in MODES:
        for k, v in opts.items():
            eng.set_option(k, v)
        tr = []
        sampler.sample(eng, tab, xT.cuda(), trace=tr)
        traces[name] = [t.cpu() for t in tr]
    del eng
    torch.cuda.empty_cache()
    return sd, doc, xT, traces


def rmse(a, b):
    return [float((x - y).pow(2).mean().sqrt()) for x, y in zip(a, b)]


out = {"steps": S, "partA_G96_vs_oracle": {}, "partB_G288_vs_split_engine": {}}
for fam, gain in (("plain", 1.0), ("tame", synth.tame_gain(S))):
    sd, doc, xT, tr = rollouts(96, gain)
    ref = []
    with torch.no_grad():
        O.Oracle(sd, 96).sample_loop(O.Schedule(S), xT, doc, trace=ref)
    out["partA_G96_vs_oracle"][fam] = {
        "last_x0_std": float(ref[-1].std()), "saturated_pixels": float((ref[-1].abs() >= 1).float().mean()),
        **{m: {"last": rmse(tr[m], ref)[-1], "every_7th": rmse(tr[m], ref)[::7]} for m, _ in MODES}}
    sd, doc, xT, tr = rollouts(288, gain)
    out["partB_G288_vs_split_engine"][fam] = {
        "last_x0_std": float(tr["split"][-1].std()),
        **{m: {"last": rmse(tr[m], tr["split"])[-1], "every_7th": rmse(tr[m], tr["split"])[::7]}
           for m in ("dither", "plain_f16")}}
print(json.dumps(out))
