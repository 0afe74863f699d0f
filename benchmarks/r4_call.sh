#!/bin/bash
mkdir -p gpurun_out/r4
(for a in 0 3 1; do echo "== r64x ablation $a"; timeout 200 python benchmarks/attn_stamps_r64m.py 16 $a r64x 2>&1 | grep -v amdgpu.ids | head -2; done
export NOSTATS=1
bash benchmarks/pmc_attn_ab.sh "r64m=" "r64x=DVD_ATTN_R64X" "x_novalu=DVD_ATTN_R64X,DVD_ATTN_R64X_ABL:1" "x_mfmaonly=DVD_ATTN_R64X,DVD_ATTN_R64X_ABL:3") 2>&1 | tee gpurun_out/r4/c39_r64x.txt
