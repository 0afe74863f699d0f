#!/bin/bash
# PMC A/B of the head_dim-256 decoder attention kernels (lab build): MFMA busy, VALU port, sustained clock, wave stalls.
# usage (GPU box): bash benchmarks/pmc_attn_ab.sh "<name>=<ENVVAR[:value][,ENVVAR...] or empty>" ...   (NOSTATS=1: counters only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_attn_ab
rm -rf $out; mkdir -p $out
export PROBE_B=16
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  for e in ${envs//,/ }; do if [[ $e == *:* ]]; then export ${e%%:*}=${e#*:}; else export $e=1; fi; done
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $out/$name -- python3 benchmarks/pmc_probe.py ${PROBE_OP:-attn256} --lab > $out/$name.log 2>&1 || tail -3 $out/$name.log
  [ -n "$NOSTATS" ] || timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${name}_stats -- python3 benchmarks/pmc_probe.py ${PROBE_OP:-attn256} --lab > $out/${name}_stats.log 2>&1 || tail -3 $out/${name}_stats.log
  for e in ${envs//,/ }; do unset ${e%%:*}; done
done
python3 - "$@" <<'PY'
import csv, glob, sys
from collections import defaultdict
for spec in sys.argv[1:]:
    name = spec.split("=")[0]
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_attn_ab/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "flash_attn" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {}
    for f in glob.glob(f"gpurun_out/pmc_attn_ab/{name}_stats/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "flash_attn" in r["Name"]:
                dur[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"]) / 1e6
    for k, d in acc.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        ms = dur.get(k)
        print(f"{name:8s} {k:40s} kernel cycles {cyc:12.0f}  MFMA busy {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):5.1f} %  "
              f"VALU port {100 * 4 * m['SQ_ACTIVE_INST_VALU'] / (cyc * 1024):4.1f} %  issue-stalled {100 * m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:4.1f} %  "
              f"parked {100 * m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:4.1f} %  un-profiled avg {ms} ms  clock(profiled cycles / un-profiled ms) "
              f"{(cyc / (ms * 1e-3) / 1e9) if ms else float('nan'):.2f} GHz")
PY
