#!/bin/bash
mkdir -p gpurun_out/r4
(timeout 600 python -m pytest tests/test_gpu_attention.py -q -x 2>&1 | tail -2
timeout 300 python benchmarks/attn_ab.py 16 7 256 "r64x=" "r64m=DVD_ATTN_R64M" 2>&1 | tail -3
timeout 200 python benchmarks/attn_stamps_r64m.py 16 0 2>&1 | grep -v amdgpu.ids | head -2) | tee gpurun_out/r4/c49_x_waits.txt
