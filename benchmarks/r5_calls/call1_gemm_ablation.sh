#!/bin/bash
# round 5, call 1: where does gemm_nt_big_kernel's time go?    -> profiles/r5_gemm_ablation.txt
O=gpurun_out/r5; mkdir -p $O
( ./benchmarks/lab/l2path_lab ) > $O/l2path_lab.txt 2>&1
( for dbg in 0 1 5 6 7 2 0; do echo "== DVD_GEMM_DEBUG=$dbg"; DVD_GEMM_DEBUG=$dbg timeout 300 python benchmarks/gemm_time.py 5 plain --lab 2>&1 | grep -v Warning; done ) > $O/gemm_ablation_wall.txt 2>&1
( DVD_GEMM_DEBUG=3 timeout 300 python benchmarks/gemm_stamps.py ) > $O/gemm_stamps.txt 2>&1
tail -50 $O/l2path_lab.txt $O/gemm_stamps.txt
