#!/bin/bash
# round 5, call 16: f16 epilogue of gemm_nt_t384_kernel with the packed words sorted by ds_bpermute (contiguous lanes per row segment)
O=gpurun_out/r5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -k "t384 or large_tile" 2>&1 | tail -2
( timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 f16 2>&1 | grep -v amdgpu.ids
  timeout 300 python benchmarks/gemm_t384_stamps.py 3072 1536 f16 2>&1 | grep -v amdgpu.ids
  for rep in 1 2; do
    echo "== t384"; timeout 300 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep TF
    echo "== 256x256"; DVD_GEMM_NO_T384=1 timeout 300 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep TF
  done ) > $O/c16_bperm.txt 2>&1
cat $O/c16_bperm.txt
