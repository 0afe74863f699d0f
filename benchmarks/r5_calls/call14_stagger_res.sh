#!/bin/bash
# round 5, call 14: the residual epilogue is an HBM burst (every CU reads + writes 393 + 393 KB at the same moment: 200 MB per tile round);
# does a start-up stagger spread it?  (quantum x 1024 cycles x 0..15 per workgroup; a tile takes ~130 k cycles)
O=gpurun_out/r5; mkdir -p $O
( for q in 0 2 4 8 0 12 4 8; do echo "== res, DVD_GEMM_T384_STAGGER=$q"; DVD_GEMM_T384_STAGGER=$q timeout 300 python benchmarks/gemm_time.py 7 res --lab 2>&1 | grep TF; done
  for q in 0 8; do echo "== stamps res, stagger $q"; DVD_GEMM_T384_STAGGER=$q timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 res 2>&1 | grep -v amdgpu.ids; done
  for q in 0 8; do echo "== f16, DVD_GEMM_T384_STAGGER=$q"; DVD_GEMM_T384_STAGGER=$q timeout 300 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep TF; done
) > $O/c14_stagger_res.txt 2>&1
cat $O/c14_stagger_res.txt
