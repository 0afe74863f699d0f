#!/bin/bash
# round 5, call 9: K-loop schedule variants of gemm_nt_t384_kernel - where the five LDS-DMA pieces sit among the phase-2 MFMAs
# (generator switch T384_PIECES, alt builds) and static priority for waves 4-7 (DVD_GEMM_T384_PRIO)
O=gpurun_out/r5; mkdir -p $O
( for rep in 1 2; do
    echo "== pieces after MFMA 1,3,5,7,9 (product)"; timeout 300 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep TF
    for t in p0 p2 p3; do echo "== alt $t"; timeout 300 python benchmarks/gemm_time.py 7 plain --lib benchmarks/lab/alt/libdvd_t384_$t.so 2>&1 | grep TF; done
    echo "== product + s_setprio 1 for waves 4-7"; DVD_GEMM_T384_PRIO=1 timeout 300 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep TF
  done ) > $O/c9_kloop_variants.txt 2>&1
cat $O/c9_kloop_variants.txt
