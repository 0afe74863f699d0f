#!/bin/bash
# round 5 call 23: what bounds the 128x128 GEMM at 2048 rows - PMC passes (never combined with a trace), then durations
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5/pmc_small
rm -rf $out; mkdir -p $out
for docs in 1 4; do
export PROBE_DOCS=$docs
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $out/mfma_$docs -- python3 benchmarks/pmc_probe.py gemm_small > $out/mfma_$docs.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $out/lds_$docs -- python3 benchmarks/pmc_probe.py gemm_small > $out/lds_$docs.log 2>&1
timeout 300 rocprofv3 --pmc TA_BUSY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/mem_$docs -- python3 benchmarks/pmc_probe.py gemm_small > $out/mem_$docs.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_$docs -- python3 benchmarks/pmc_probe.py gemm_small > $out/fetch_$docs.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$docs -- python3 benchmarks/pmc_probe.py gemm_small > $out/stats_$docs.log 2>&1
done
python3 - <<'P'
import csv, glob, os, collections
out = "gpurun_out/r5/pmc_small"
for docs in (1, 4):
    print(f"== gemm_small, {2048 * docs} rows x 1536 x 1536, (hi, lo)")
    for grp in ("mfma", "lds", "mem", "fetch"):
        acc = collections.defaultdict(list)
        for f in glob.glob(f"{out}/{grp}_{docs}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_nt" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"  {grp:5s} {k:28s} launches {len(v):3d}  mean {sum(v) / len(v):16.1f}")
    for f in glob.glob(f"{out}/stats_{docs}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_nt" in r["Name"]:
                print(f"  stats {r['Name'][:60]} calls {r['Calls']} avg {float(r['AverageNs']) / 1e3:.1f} us")
P
