#!/usr/bin/env python3
"""Fold the FETCH_SIZE / WRITE_SIZE passes of benchmarks/pmc_traffic.sh into profiles/<round>_pmc_traffic.json.
usage: python benchmarks/pmc_traffic_json.py gpurun_out/pmc_traffic profiles/archive/r1_pmc_traffic.json
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests
at 64 bytes (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact for 16-byte-per-lane stores."""
import csv
import glob
import json
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dvd" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        fe, wr = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        out[k] = {"FETCH_SIZE_KiB": fe, "WRITE_SIZE_KiB": wr, "launches": len(d["FETCH_SIZE"]),
                  "hbm_bytes_per_launch": (2 * fe + wr) * 1024}
out["_how"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 benchmarks/pmc_probe.py <op> with "
               "PROBE_B=16 (the bench's launch shape: 16 samples x 6 heads, T = 20736); bytes = (2*FETCH + WRITE) KiB")
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
