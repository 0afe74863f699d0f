#!/bin/bash
mkdir -p gpurun_out/r4
(python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep -v amdgpu; DVD_GEMM_M16=1 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep -v amdgpu; python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep -v amdgpu) | tee gpurun_out/r4/c59_gemm_m16.txt
