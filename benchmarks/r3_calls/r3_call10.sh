#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c10
export TMPDIR=/tmp
{
for B in 16 64; do
python benchmarks/attn_time.py 64 $B 7 2>&1 | grep -v amdgpu
python benchmarks/attn_time.py 64 $B 7 --lib benchmarks/lab/alt/libdvd_hip_occ3.so 2>&1 | grep -v amdgpu
python benchmarks/attn_time.py 64 $B 7 2>&1 | grep -v amdgpu
python benchmarks/attn_time.py 64 $B 7 --lib benchmarks/lab/alt/libdvd_hip_occ3.so 2>&1 | grep -v amdgpu
done
} > gpurun_out/c10/attn64_occ.txt; cat gpurun_out/c10/attn64_occ.txt
python -m pytest tests/test_gpu_engine.py -q -m gpu -s -k "tame_family_loop_vs_reference" 2>&1 | grep -E "tame-family|passed|failed"
