#!/bin/bash
mkdir -p gpurun_out/r4
(timeout 300 python benchmarks/attn_ab.py 16 7 256 "r64m=" "r64x=DVD_ATTN_R64X" 2>&1 | tail -3
export NOSTATS=1
bash benchmarks/pmc_attn_ab.sh "r64m=" "r64x=DVD_ATTN_R64X") 2>&1 | tee gpurun_out/r4/c45_x_vs_m.txt
