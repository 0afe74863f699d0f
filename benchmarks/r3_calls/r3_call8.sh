#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c8
export TMPDIR=/tmp
{
python benchmarks/gemm_time.py 9 plain 2>&1 | grep -v amdgpu
DVD_GEMM_M32=1 python benchmarks/gemm_time.py 9 plain --lab 2>&1 | grep -v amdgpu
python benchmarks/gemm_time.py 9 plain 2>&1 | grep -v amdgpu
DVD_GEMM_M32=1 python benchmarks/gemm_time.py 9 plain --lab 2>&1 | grep -v amdgpu
} > gpurun_out/c8/gemm_time.txt; cat gpurun_out/c8/gemm_time.txt
