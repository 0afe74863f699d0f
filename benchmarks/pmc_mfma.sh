#!/bin/bash
# MFMA utilisation of the three dominant kernels from PMC counters (bench launch shapes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_mfma
rm -rf $out; mkdir -p $out
export PROBE_B=16
for op in attn256 attn64 gemm_split; do
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $out/$op -- python3 benchmarks/pmc_probe.py $op > $out/$op.log 2>&1 || tail -3 $out/$op.log
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/pmc_mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dvd" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
    util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)   # 256 CUs x 4 SIMDs
    print(f"{k:44s} launches {len(d['GRBM_GUI_ACTIVE'])}  kernel cycles {cyc:12.0f}  MFMA busy {100 * util:5.1f} % of SIMD-cycles  "
          f"waves: issue-stalled {100 * m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:4.1f} %  parked {100 * m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:4.1f} %  "
          f"VALU port {100 * 4 * m['SQ_ACTIVE_INST_VALU'] / (cyc * 1024):4.1f} %")
PY
