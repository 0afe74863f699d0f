#!/bin/bash
# PMC passes for the drop-in grid_sample (8 documents per launch = the bench's roofline_unwarp leg):
#   lds  = the product's LDS-tile kernel        rows = round 2's row kernel (lab build, DVD_WARP_NOLDS=1)
# FETCH_SIZE / WRITE_SIZE in separate passes; TCC hit/miss; SQ occupancy/stall counters; a --stats pass for durations.
out=${1:-gpurun_out/pmc_warp}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf $out; mkdir -p $out
run() {  # tag, extra args...
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_$c -- python3 benchmarks/pmc_probe.py "$@" > $out/${tag}_$c.log 2>&1
  done
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/${tag}_tcc -- python3 benchmarks/pmc_probe.py "$@" > $out/${tag}_tcc.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_sq -- python3 benchmarks/pmc_probe.py "$@" > $out/${tag}_sq.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 benchmarks/pmc_probe.py "$@" > $out/${tag}_stats.log 2>&1
}
run lds gridsample8
export DVD_WARP_NOLDS=1
run rows gridsample8 --lab
unset DVD_WARP_NOLDS
run copy copy8
python3 - "$out" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for tag in ("lds", "rows", "copy"):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, tag + "_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "grid_sample" in k or "copy" in k.lower() or "elementwise" in k:
                acc[(k.split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    dur = {}
    for f in glob.glob(os.path.join(out, tag + "_stats", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Name"].split("(")[0][-60:]] = (float(r["AverageNs"]) * 1e-6, r["Calls"])
    print("==", tag)
    for k, v in sorted(dur.items()):
        if "grid_sample" in k or "copy" in k.lower() or "elementwise" in k:
            print(f"   {k}: avg {v[0]:.4f} ms x {v[1]}")
    for (k, c), v in sorted(acc.items()):
        print(f"   {k} {c}: mean {sum(v) / len(v):.4g} (n={len(v)})")
PY
