#!/bin/bash
# round-3 GPU call 1: new parity tests + dither on/off bench + kernel profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
python -m pytest tests/test_gpu_gemm.py -q -m gpu -x -k "dither" -s > gpurun_out/c1/t_dither.log 2>&1; echo "dither tests rc=$?"
python -m pytest tests/test_gpu_engine.py -q -m gpu -s > gpurun_out/c1/t_engine.log 2>&1; echo "engine tests rc=$?"
python -m pytest tests/test_gpu_dropin.py -q -m gpu -s -k "checkpoint_files or without_prestage" > gpurun_out/c1/t_dropin.log 2>&1; echo "dropin tests rc=$?"
python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/c1/bench_dither.json 2> gpurun_out/c1/bench_dither.err; echo "bench dither rc=$?"
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dither > gpurun_out/c1/bench_split.json 2> gpurun_out/c1/bench_split.err; echo "bench split rc=$?"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c1/prof" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/c1/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/c1/prof_bench.err"; echo "prof rc=$?"
cd "$GRAFT_REPO_ROOT"
python benchmarks/stats_summary.py "$(find gpurun_out/c1/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/c1/prof_bench.json "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (round 3, dithered weights)" > gpurun_out/c1/prof_summary.txt 2>&1
tail -3 gpurun_out/c1/t_dither.log; grep -E "passed|failed|error" gpurun_out/c1/t_engine.log | tail -3; tail -3 gpurun_out/c1/t_dropin.log
cat gpurun_out/c1/bench_dither.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dither', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
cat gpurun_out/c1/bench_split.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
head -20 gpurun_out/c1/prof_summary.txt
