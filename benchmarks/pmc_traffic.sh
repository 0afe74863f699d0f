#!/bin/bash
# HBM traffic per launch for the dominant kernels, one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass).
# run on the GPU box from the repo root:  bash benchmarks/pmc_traffic.sh   -> gpurun_out/pmc_traffic/<op>_<counter>/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_traffic
mkdir -p $out
export PROBE_B=16
for op in attn256 attn64 gemm unwarp; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/${op}_$c -- python3 benchmarks/pmc_probe.py $op > $out/${op}_$c.log 2>&1
  done
done
python3 benchmarks/pmc_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
