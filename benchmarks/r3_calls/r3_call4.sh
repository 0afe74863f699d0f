#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c4
rocprofv3 -L > gpurun_out/c4/counters.txt 2>&1 || rocprofv3-avail list > gpurun_out/c4/counters.txt 2>&1
bash benchmarks/pmc_warp.sh gpurun_out/c4/pmc_warp > gpurun_out/c4/pmc_warp_summary.txt 2>&1
cat gpurun_out/c4/pmc_warp_summary.txt
grep -c . gpurun_out/c4/counters.txt
