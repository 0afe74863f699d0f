#!/usr/bin/env python3
"""(Checker-side experiment: lives under tests/ because it uses the oracle.)  The headline configuration's WHOLE loop
against the CPU oracle: 50-step DDIM at G = 288 (T = 20 736 tokens), one document, one hypothesis - every step's x0
prediction and the final clamped map.  The oracle needs ~1 minute per evaluation on 32 host threads, so this is run once
per round on the GPU box (not a pytest case) and its output is committed under profiles/.
Round 3: the weight family is selectable (tame = out_gain 1.6/steps: x0 stays inside (-1, 1), nothing hides behind the
final clamp) and ONE oracle run serves three engine runs - dithered weights (the default at this grid), the (hi, lo) split
in every GEMM (round 2's default) and plain f16.
The oracle needs an hour of host time and no GPU, the engine a minute of GPU time: the two halves can run apart -
  python tests/tools/parity_g288.py 50 288 tame --engine-only traces.npz     (GPU box: the three engine roll-outs, every 7th
                                                                              step's x0 + the last + the final map)
  python tests/tools/parity_g288.py 50 288 tame --oracle-only traces.npz     (any host: the oracle roll-out, compared)
usage: python tests/tools/parity_g288.py [steps=50] [grid=288] [family=tame|plain] [--engine-only F | --oracle-only F]
       > profiles/<round>_parity_g288.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dvd_amd import schedule, synth  # noqa: E402
from oracle import dvd_oracle as O  # noqa: E402

engine_only = sys.argv[sys.argv.index("--engine-only") + 1] if "--engine-only" in sys.argv else None
oracle_only = sys.argv[sys.argv.index("--oracle-only") + 1] if "--oracle-only" in sys.argv else None
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 288
family = sys.argv[3] if len(sys.argv) > 3 else "tame"
torch.set_num_threads(min(int(os.environ.get("ORACLE_THREADS", "32")), os.cpu_count() or 8))
sd = synth.synth_state_dict(grid, 7, blocks=[11], out_gain=synth.tame_gain(steps) if family == "tame" else 1.0)
d0 = synth.synth_document(0, grid, 1234)
keys = ("y512", "mask_cat", "mask_y512", "line_msk")
doc = {k: torch.from_numpy(d0[k])[None] for k in keys}
xT = torch.from_numpy(synth.synth_noise(0, 1, grid, 1234))
tab = schedule.Tables(schedule.named_betas("cosine", steps))
KEEP = sorted(set(range(0, steps, 7)) | {steps - 1})          # the steps whose x0 travels between the two halves
runs = {}
if oracle_only is None:
    from dvd_amd import sampler
    from dvd_amd.engine import Engine
    eng = Engine(grid, 1, 1)
    eng.load_state_dict(sd)
    eng.prepare(*(doc[k].cuda() for k in keys))
    for name, opts in (("dither", {"dither": 1, "split_weights": 1}), ("split", {"dither": 0, "split_weights": 1}),
                       ("plain_f16", {"dither": 0, "split_weights": 0})):
        for k, v in opts.items():
            eng.set_option(k, v)
        tr = []
        out = sampler.sample(eng, tab, xT.cuda(), trace=tr)
        runs[name] = ({i: tr[i].cpu() for i in (KEEP if engine_only else range(steps))}, out.cpu())
    del eng
    torch.cuda.empty_cache()
    if engine_only:
        np.savez_compressed(engine_only, **{f"{n}/x0_{i}": t.numpy() for n, (tr, _) in runs.items() for i, t in tr.items()},
                            **{f"{n}/final": o.numpy() for n, (_, o) in runs.items()})
        print(json.dumps({"engine_traces": engine_only, "steps_kept": KEEP}))
        sys.exit(0)
else:
    z = np.load(oracle_only)
    for n in ("dither", "split", "plain_f16"):
        runs[n] = ({i: torch.from_numpy(z[f"{n}/x0_{i}"]) for i in KEEP}, torch.from_numpy(z[f"{n}/final"]))
orc = O.Oracle(sd, grid)
tr_ref = []
t0 = time.time()
with torch.no_grad():
    ref = orc.sample_loop(O.Schedule(steps), xT, doc, trace=tr_ref)
dt = time.time() - t0
res = {"what": f"{steps}-step DDIM at G={grid}, 1 document x 1 hypothesis, synthetic weights (seed 7, {family} family): HIP "
               "engine vs CPU oracle; coordinate RMSE of the final clamped map and of every step's UN-CLAMPED x0",
       "last_x0_std": float(tr_ref[-1].std()), "saturated_pixels_last_x0": float((tr_ref[-1].abs() >= 1).float().mean()),
       "bar": 1e-3, "oracle_seconds": round(dt, 1), "oracle_threads": torch.get_num_threads(),
       "t_model": [tab.model_time(i) for i in range(steps - 1, -1, -1)], "weights": {}}
for name, (tr, out) in runs.items():
    per = {i: float((tr[i] - tr_ref[i]).pow(2).mean().sqrt()) for i in sorted(tr)}
    res["weights"][name] = {"final_coord_rmse": float((out - ref).pow(2).mean().sqrt()),
                            "final_max_abs": float((out - ref).abs().max()), "per_step_x0_rmse": per,
                            "ok": bool(per[steps - 1] < 1e-3 and float((out - ref).pow(2).mean().sqrt()) < 1e-3)}
print(json.dumps(res))
