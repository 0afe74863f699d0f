#!/bin/bash
# after the dwconv tile kernel: smoke, token / engine-forward tests, kernel stats at HEAD, one bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/final2
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m pytest tests/test_gpu_tokens.py tests/test_gpu_dropin.py -q -m gpu 2>&1 | tail -2
python -m pytest tests/test_gpu_engine.py -q -m gpu -k "forward or golden or batched or graph" 2>&1 | tail -2
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/final2/prof" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > "$GRAFT_REPO_ROOT/gpurun_out/final2/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/final2/prof_bench.err"; echo "prof rc=$?"
cd "$GRAFT_REPO_ROOT"
python benchmarks/stats_summary.py "$(find gpurun_out/final2/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/final2/prof_bench.json "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs (round 3, final HEAD)" > gpurun_out/final2/prof_summary.txt 2>&1
sed -n 3,12p gpurun_out/final2/prof_summary.txt | cut -c1-130
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/final2/bench.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/final2/bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
