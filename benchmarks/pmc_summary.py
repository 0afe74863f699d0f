#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel, mean of each counter over dispatches."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "dvd" not in name:
            continue
        acc[name[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} n={len(v):3d} mean={sum(v) / len(v):16.1f}")
