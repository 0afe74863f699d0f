#!/usr/bin/env python3
"""(Checker-side experiment: lives under tests/ because it uses the oracle.)  The headline configuration's WHOLE loop
against the CPU oracle: 50-step DDIM at G = 288 (T = 20 736 tokens), one document, one hypothesis - every step's x0
prediction and the final clamped map.  The oracle needs ~1 minute per evaluation on 32 host threads, so this is run once
per round on the GPU box (not a pytest case) and its output is committed under profiles/.
usage: python tests/tools/parity_g288.py [steps=50] [grid=288] > profiles/<round>_parity_g288.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dvd_amd import sampler, schedule, synth  # noqa: E402
from dvd_amd.engine import Engine  # noqa: E402
from oracle import dvd_oracle as O  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 288
torch.set_num_threads(min(32, os.cpu_count() or 8))
sd = synth.synth_state_dict(grid, 7, blocks=[11])
d0 = synth.synth_document(0, grid, 1234)
keys = ("y512", "mask_cat", "mask_y512", "line_msk")
doc = {k: torch.from_numpy(d0[k])[None] for k in keys}
xT = torch.from_numpy(synth.synth_noise(0, 1, grid, 1234))
eng = Engine(grid, 1, 1)
eng.load_state_dict(sd)
eng.prepare(*(doc[k].cuda() for k in keys))
tab = schedule.Tables(schedule.named_betas("cosine", steps))
tr = []
out = sampler.sample(eng, tab, xT.cuda(), trace=tr)
tr = [t.cpu() for t in tr]
out = out.cpu()
del eng
torch.cuda.empty_cache()
orc = O.Oracle(sd, grid)
tr_ref = []
t0 = time.time()
with torch.no_grad():
    ref = orc.sample_loop(O.Schedule(steps), xT, doc, trace=tr_ref)
dt = time.time() - t0
per = [float((a - b).pow(2).mean().sqrt()) for a, b in zip(tr, tr_ref)]
res = {"what": f"{steps}-step DDIM at G={grid}, 1 document x 1 hypothesis, synthetic weights (seed 7): HIP engine vs CPU oracle",
       "final_coord_rmse": float((out - ref).pow(2).mean().sqrt()), "final_max_abs": float((out - ref).abs().max()),
       "per_step_x0_rmse": per, "bar": 1e-3, "ok": bool(float((out - ref).pow(2).mean().sqrt()) < 1e-3),
       "oracle_seconds": round(dt, 1), "oracle_threads": torch.get_num_threads(),
       "t_model": [tab.model_time(i) for i in range(steps - 1, -1, -1)]}
print(json.dumps(res))
