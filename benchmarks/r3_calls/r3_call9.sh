#!/bin/bash
# round-3 GPU call 9: whole GPU suite (durations), driver-like bench, rocprofv3 kernel stats, PMC round
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c9
export TMPDIR=/tmp
( time python -m pytest tests -q -m gpu --durations=25 ) > gpurun_out/c9/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|^real" gpurun_out/c9/gpu_tests.log | tail -3
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/c9/bench_driverlike.json 2> gpurun_out/c9/bench_driverlike.err; echo "bench rc=$?"
grep real gpurun_out/c9/bench_driverlike.err
python -c "
import json; d=json.loads(open('gpurun_out/c9/bench_driverlike.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_unwarp']['achieved'], d['cpu_baseline']['value'])"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c9/prof" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > "$GRAFT_REPO_ROOT/gpurun_out/c9/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/c9/prof_bench.err"; echo "prof rc=$?"
cd "$GRAFT_REPO_ROOT"
python benchmarks/stats_summary.py "$(find gpurun_out/c9/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/c9/prof_bench.json "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs (round 3 final kernels)" > gpurun_out/c9/prof_summary.txt 2>&1
head -14 gpurun_out/c9/prof_summary.txt | cut -c1-150
bash benchmarks/pmc_round.sh r3 > gpurun_out/c9/pmc_round.log 2>&1; echo "pmc rc=$?"
tail -12 gpurun_out/c9/pmc_round.log
