#!/usr/bin/env python3
"""Runs bench.py's configs[3] leg alone at 4 documents (its fallback size) to exercise the leg's parity code: the fused DDPM step vs
the oracle and document 0 of the batch == the same document alone after all 250 ancestral steps.  usage: python benchmarks/cfg3_leg_check.py"""
import json, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench
from dvd_amd import synth
from dvd_amd.engine import Engine, aligned_empty
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
eng = Engine(288, 1, 2, device=dev)
_, nbytes = eng.blob_layout()
blob = aligned_empty(nbytes, dev)
blob.copy_(eng.pack_blob(synth.synth_state_dict(288, seed=7, blocks=[11])))
del eng
bench.CFG3_FULL_DEADLINE_S = -1          # -> the leg runs its 4-document fallback
out = bench.other_configs(dev, blob, 2, 3508, 2480, want_cpu=True, legs=("cfg3",))
print(json.dumps(out, indent=1))
