#!/bin/bash
# round-3 GPU call 7: 16x16x32 GEMM (tests + timing A/B + bench), batched pre-stage (tests + timing)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c7
export TMPDIR=/tmp
python -m pytest tests/test_gpu_gemm.py -q -m gpu -x > gpurun_out/c7/t_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -3 gpurun_out/c7/t_gemm.log
python -m pytest tests/test_gpu_prestage.py -q -m gpu -x > gpurun_out/c7/t_prestage.log 2>&1; echo "prestage tests rc=$?"; tail -3 gpurun_out/c7/t_prestage.log
{
python benchmarks/gemm_time.py 7 plain 2>/dev/null
DVD_GEMM_M32=1 python benchmarks/gemm_time.py 7 plain --lab 2>/dev/null
python benchmarks/gemm_time.py 7 split 2>/dev/null
} > gpurun_out/c7/gemm_time.txt; cat gpurun_out/c7/gemm_time.txt
{
python benchmarks/prestage_time.py 64 1 2>/dev/null
python benchmarks/prestage_time.py 64 8 2>/dev/null
python benchmarks/prestage_time.py 64 16 2>/dev/null
} > gpurun_out/c7/prestage_time.txt; cat gpurun_out/c7/prestage_time.txt
python -m pytest tests/test_gpu_engine.py -q -m gpu -x -k "forward or golden or batched or long_loop" > gpurun_out/c7/t_engine.log 2>&1; echo "engine tests rc=$?"; tail -3 gpurun_out/c7/t_engine.log
python tests/tools/parity_g288.py 50 288 tame --engine-only gpurun_out/c7/g288_engine_traces.npz > gpurun_out/c7/g288_engine.log 2>&1; echo "traces rc=$?"; ls -la gpurun_out/c7/g288_engine_traces.npz
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/c7/bench_m16.json 2> gpurun_out/c7/bench_m16.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/c7/bench_m16.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_unwarp']['achieved'])"
