#!/bin/bash
# round-3 GPU call 12: every per-step GEMM weight dithered (256 x 128 kernel on one weight tensor): tests, parity loops, profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c12
export TMPDIR=/tmp
python -m pytest tests/test_gpu_gemm.py -q -m gpu -x > gpurun_out/c12/t_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -2 gpurun_out/c12/t_gemm.log
python -m pytest tests/test_gpu_engine.py -q -m gpu -x -s -k "long_loop or ddpm_large or baseline_grid or graph_replay or batched" > gpurun_out/c12/t_engine.log 2>&1; echo "engine tests rc=$?"
grep -E "long loop rmse|split \(dither|plain f16|ddpm|G=288|passed|failed" gpurun_out/c12/t_engine.log | cut -c1-260
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c12/prof" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > "$GRAFT_REPO_ROOT/gpurun_out/c12/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/c12/prof_bench.err"; echo "prof rc=$?"
cd "$GRAFT_REPO_ROOT"
python benchmarks/stats_summary.py "$(find gpurun_out/c12/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/c12/prof_bench.json "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs (round 3: every per-step GEMM weight dithered)" > gpurun_out/c12/prof_summary.txt 2>&1
sed -n 3,14p gpurun_out/c12/prof_summary.txt | cut -c1-130
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/c12/bench.json 2> gpurun_out/c12/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/c12/bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
