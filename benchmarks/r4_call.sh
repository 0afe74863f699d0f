#!/bin/bash
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_gemm_ab; rm -rf $out; mkdir -p $out
for v in base m16; do
  [ $v = m16 ] && export DVD_GEMM_M16=1
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $out/$v -- python3 benchmarks/pmc_probe.py gemm --lab > $out/$v.log 2>&1 || tail -3 $out/$v.log
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${v}_stats -- python3 benchmarks/pmc_probe.py gemm --lab > $out/${v}_stats.log 2>&1
done
python3 - <<'PY' | tee gpurun_out/r4/c70_gemm_pmc.txt
import csv, glob
from collections import defaultdict
for name in ("base", "m16"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_gemm_ab/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {}
    for f in glob.glob(f"gpurun_out/pmc_gemm_ab/{name}_stats/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Name"]:
                dur[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"]) / 1e6
    for k, d in acc.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        ms = dur.get(k)
        print(f"{name:5s} {k:36s} cycles {cyc:10.0f}  MFMA busy {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):5.1f} %  issue-stalled {100 * m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:4.1f} %  parked {100 * m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:4.1f} %  avg {ms} ms  clock {(cyc / (ms * 1e-3) / 1e9) if ms else float('nan'):.2f} GHz")
PY
