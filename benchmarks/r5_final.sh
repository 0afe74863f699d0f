#!/bin/bash
# round-5 evidence at HEAD: smoke(), whole GPU suite, the driver's bench invocation, rocprofv3 kernel stats, PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/final5
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final5/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/final5/smoke.log
( time python -m pytest tests -q -m gpu --durations=12 ) > gpurun_out/final5/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|^real" gpurun_out/final5/gpu_tests.log | tail -3
bash benchmarks/pmc_round.sh r5 > gpurun_out/final5/pmc_round.log 2>&1; echo "pmc rc=$?"
cp gpurun_out/pmc_r5/r5_pmc_*.json profiles/ 2>/dev/null
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/final5/bench_driverlike.json 2> gpurun_out/final5/bench_driverlike.err; echo "bench rc=$?"
grep real gpurun_out/final5/bench_driverlike.err
python -c "
import json; d=json.loads(open('gpurun_out/final5/bench_driverlike.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline_unwarp']['achieved'], d['roofline_unwarp']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores']); print({k: (v['value'], v['parity']['ok'] if v.get('parity') else None) for k, v in d['other_configs'].items()})"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/final5/prof" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > "$GRAFT_REPO_ROOT/gpurun_out/final5/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/final5/prof_bench.err"; echo "prof rc=$?"
cd "$GRAFT_REPO_ROOT"
python benchmarks/stats_summary.py "$(find gpurun_out/final5/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/final5/prof_bench.json "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs (round 5)" > gpurun_out/final5/prof_summary.txt 2>&1
sed -n 3,12p gpurun_out/final5/prof_summary.txt | cut -c1-130
