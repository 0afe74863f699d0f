#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5
python benchmarks/gemm_l2warm.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/call36.txt; cat gpurun_out/r5/call36.txt
