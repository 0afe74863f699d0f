#!/bin/bash
mkdir -p gpurun_out/r4
(timeout 900 python -m pytest tests/test_gpu_attention.py -q 2>&1 | tail -3
timeout 900 python benchmarks/attn_lib_ab.py 64 benchmarks/lab/libdvd_hip_prev.so dvd_amd/libdvd_hip.so 3 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/r4/c68_h64x_negm.txt
