#!/bin/bash
mkdir -p gpurun_out/r4
for a in 0; do echo "== ablation $a"; timeout 200 python benchmarks/attn_stamps_r64m.py 16 $a 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r4/c24_stamps.txt
