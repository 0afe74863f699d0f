#!/usr/bin/env python3
"""The reference's native operating point (G = 64, 3-step DDIM, 2 hypotheses, from decoded images), one document at a time
and batched: bench.py's `native_point*` legs alone (development loop; the driver's line carries the same legs).
usage: python benchmarks/native_point.py [--no-cpu]"""
import importlib.util, json, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
torch.cuda.set_device(0)
out = bench.other_configs(torch.device("cuda", 0), None, 2, 3508, 2480, want_cpu="--no-cpu" not in sys.argv, legs=("native",))
print(json.dumps(out, indent=1))
