#!/usr/bin/env python3
"""Fold the passes of benchmarks/pmc_round.sh into <dst>/<round>_pmc_traffic.json, <round>_pmc_mfma.json and a text summary.
Keys are the kernel names exactly as dvd_flash_attn_kernel_name() / rocprofv3 report them (without 'void ', 'dvd::' and the
argument list), each record carries the launch shape it was measured at: bench.py attaches a record to its roofline only
when BOTH match the kernel it launched.
  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB: both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at
  64 bytes (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); clock = kernel cycles / kernel duration.
usage: python benchmarks/pmc_round_json.py <pmc dir> <dst dir> <round>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, dst, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
SHAPE = {"samples": int(os.environ.get("PROBE_B", "16")), "grid": 288}
WARP_SHAPE = {"documents": 8, "h": 3508, "w": 2480}          # benchmarks/pmc_probe.py gridsample8


def shape_of(kernel):
    return WARP_SHAPE if ("grid_sample" in kernel or "unwarp" in kernel) else SHAPE


def clean(name):
    return name.split("(")[0].replace("void ", "").replace("dvd::", "").strip()


def counters(pattern):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(src, pattern, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "dvd" in r["Kernel_Name"]:
                acc[clean(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def durations():
    out = {}
    for f in glob.glob(os.path.join(src, "stats_*", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "dvd" in r["Name"]:
                out[clean(r["Name"])] = float(r["AverageNs"]) * 1e-6
    return out


mean = lambda v: sum(v) / len(v)  # noqa: E731
dur = durations()
traffic = {}
for k, d in counters("traffic_*").items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        fe, wr = mean(d["FETCH_SIZE"]), mean(d["WRITE_SIZE"])
        traffic[k] = {"FETCH_SIZE_KiB": fe, "WRITE_SIZE_KiB": wr, "launches": len(d["FETCH_SIZE"]),
                      "hbm_bytes_per_launch": (2 * fe + wr) * 1024, "launch_shape": shape_of(k), "kernel_ms": dur.get(k)}
traffic["_how"] = ("benchmarks/pmc_round.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over "
                   "benchmarks/pmc_probe.py <op> with PROBE_B=16 (the bench's launch shape: 16 samples, T = 20736); "
                   "bytes = (2*FETCH + WRITE) KiB (gfx950 FETCH_SIZE correction)")
mfma = {}
for k, d in counters("mfma_*").items():
    if "GRBM_GUI_ACTIVE" not in d:
        continue
    m = {c: mean(v) for c, v in d.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    rec = {"mfma_busy": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4),
           "valu_port_busy": round(4 * m["SQ_ACTIVE_INST_VALU"] / (cyc * 1024), 4),
           "waves_issue_stalled": round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 4),
           "kernel_cycles": int(cyc), "kernel_ms": dur.get(k), "launch_shape": shape_of(k)}
    if dur.get(k):
        rec["sustained_clock_ghz"] = round(cyc / (dur[k] * 1e-3) / 1e9, 3)
    mfma[k] = rec
mfma["_how"] = ("benchmarks/pmc_round.sh: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY "
                "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE; durations from a separate --kernel-trace "
                "--stats pass of the same probe; mfma_busy = MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)")
json.dump(traffic, open(os.path.join(dst, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
json.dump(mfma, open(os.path.join(dst, f"{rnd}_pmc_mfma.json"), "w"), indent=1)
with open(os.path.join(dst, f"{rnd}_pmc_summary.txt"), "w") as f:
    for k in sorted(set(traffic) | set(mfma)):
        if k.startswith("_"):
            continue
        t, m = traffic.get(k, {}), mfma.get(k, {})
        f.write(f"{k:46s} ms {dur.get(k, float('nan')):9.3f}  HBM/launch {t.get('hbm_bytes_per_launch', float('nan')) / 1e6:10.1f} MB  "
                f"MFMA busy {100 * m.get('mfma_busy', float('nan')):5.1f} %  VALU port {100 * m.get('valu_port_busy', float('nan')):5.1f} %  "
                f"clock {m.get('sustained_clock_ghz', float('nan')):5.2f} GHz\n")
print(open(os.path.join(dst, f"{rnd}_pmc_summary.txt")).read())
