#!/usr/bin/env python3
"""(Checker-side experiment: lives under tests/ because it uses the oracle.)  Which f16 weights need the hi + 2^-11 lo split?  Zeroes the `_lo` half of one weight group at a time and measures
the 50-step DDIM coordinate RMSE against the CPU oracle (same harness as tests/test_gpu_engine.py).
usage: python tests/tools/lo_ablation.py [grid=32] [steps=50]"""
import os
import re
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dvd_amd import engine as E, sampler, schedule, synth, weights  # noqa: E402
from oracle import dvd_oracle as O  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
torch.set_num_threads(min(32, os.cpu_count() or 8))
sd = synth.synth_state_dict(grid, 7, blocks=[11])
d0 = synth.synth_document(0, grid, 1234)
doc = {k: torch.from_numpy(d0[k])[None] for k in ("y512", "mask_cat", "mask_y512", "line_msk")}
xT = torch.from_numpy(synth.synth_noise(0, 2, grid, 1234))
orc = O.Oracle(sd, grid)
tr_ref = []
ref = orc.sample_loop(O.Schedule(steps), xT, doc, trace=tr_ref)
tab = schedule.Tables(schedule.named_betas("cosine", steps))

GROUPS = {
    "full split (baseline)": r"^$",
    "decoder FFN lo = 0 (c1w16, c2w16)": r"^d\d_c[12]w16_lo$",
    "decoder attention lo = 0 (wqk16, wv16, wfc16)": r"^d\d_w(qk|v|fc)16_lo$",
    "everything except decoder lo = 0": r"^(?!d\d_).*16_lo$",
    "decoder FFN + non-decoder lo = 0": r"^(d\d_c[12]w16_lo|(?!d\d_).*16_lo)$",
    "no split at all": r".*16_lo$",
}
orig_pack = weights.pack
for name, pat in GROUPS.items():
    rx = re.compile(pat)

    def pack(state_dict, g, _rx=rx):
        out = orig_pack(state_dict, g)
        hit = [k for k in out if _rx.match(k)]
        for k in hit:
            out[k] = torch.zeros_like(out[k])
        pack.hit = len(hit)
        return out
    weights.pack = pack
    eng = E.Engine(grid, 1, 2)
    eng.load_state_dict(sd)
    eng.prepare(*(doc[k].cuda() for k in ("y512", "mask_cat", "mask_y512", "line_msk")))
    tr = []
    out = sampler.sample(eng, tab, xT.cuda(), trace=tr)
    per = [float((a.cpu() - b).pow(2).mean().sqrt()) for a, b in zip(tr, tr_ref)]
    err = float((out.cpu() - ref).pow(2).mean().sqrt())
    print(f"G={grid} S={steps}  {name:48s} zeroed {pack.hit:3d} tensors   final RMSE {err:.3e}   per-step x0 RMSE last {per[-1]:.3e} max {max(per):.3e}", flush=True)
    del eng
weights.pack = orig_pack
