#!/bin/bash
# round 5, call 5: start-up stagger of gemm_nt_t384_kernel (quantum x 1024 cycles x 0..15)
O=gpurun_out/r5; mkdir -p $O
( for q in 0 1 2 4 0 8 1 2; do echo "== DVD_GEMM_T384_STAGGER=$q"; DVD_GEMM_T384_STAGGER=$q timeout 300 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep TF; done ) > $O/c5_stagger.txt 2>&1
cat $O/c5_stagger.txt
( for q in 0 2; do echo "== stamps, stagger $q"; DVD_GEMM_T384_STAGGER=$q timeout 300 python benchmarks/gemm_t384_stamps.py 3072 1536; done ) > $O/c5_stagger_stamps.txt 2>&1
cat $O/c5_stagger_stamps.txt
