#!/bin/bash
# round 5, call 10: are streaming (nt) stores acknowledged sooner?  residual flavour of gemm_nt_t384_kernel, stamps + wall
O=gpurun_out/r5; mkdir -p $O
( for rep in 1 2; do
    echo "== res, plain stores"; timeout 300 python benchmarks/gemm_time.py 5 res --lab 2>&1 | grep TF
    echo "== res, nt stores"; DVD_GEMM_T384_NT=1 timeout 300 python benchmarks/gemm_time.py 5 res --lab 2>&1 | grep TF
  done
  echo "== stamps plain"; timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 res 2>&1 | grep -v amdgpu.ids
  echo "== stamps nt"; DVD_GEMM_T384_NT=1 timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 res 2>&1 | grep -v amdgpu.ids
) > $O/c10_nt_res.txt 2>&1
cat $O/c10_nt_res.txt
