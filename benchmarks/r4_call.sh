#!/bin/bash
mkdir -p gpurun_out/r4
(timeout 300 python -m pytest tests/test_gpu_attention.py -q -x -k "r64x or exit" 2>&1 | tail -2
timeout 300 python benchmarks/attn_ab.py 16 7 256 "r64m=" "r64x=DVD_ATTN_R64X" 2>&1 | tail -3
timeout 200 python benchmarks/attn_stamps_r64m.py 16 0 r64x 2>&1 | grep -v amdgpu.ids | head -1) | tee gpurun_out/r4/c47_x_fma.txt
