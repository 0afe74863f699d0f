#!/bin/bash
# round 5, call 13: the phased residual epilogue of gemm_nt_t384_kernel against the interleaved one (lab switch) and the 256 x 256 kernel
O=gpurun_out/r5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -k "t384" 2>&1 | tail -2
( for rep in 1 2; do
    echo "== t384 res, phased"; timeout 300 python benchmarks/gemm_time.py 5 res --lab 2>&1 | grep TF
    echo "== t384 res, interleaved"; DVD_GEMM_T384_RES_INTERLEAVED=1 timeout 300 python benchmarks/gemm_time.py 5 res --lab 2>&1 | grep TF
    echo "== 256x256 res"; DVD_GEMM_NO_T384=1 timeout 300 python benchmarks/gemm_time.py 5 res --lab 2>&1 | grep TF
  done
  echo "== stamps phased"; timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 res 2>&1 | grep -v amdgpu.ids
  echo "== stamps interleaved"; DVD_GEMM_T384_RES_INTERLEAVED=1 timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 res 2>&1 | grep -v amdgpu.ids
  echo "== stamps f16"; timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 f16 2>&1 | grep -v amdgpu.ids
) > $O/c13_phased.txt 2>&1
cat $O/c13_phased.txt
