#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5
python benchmarks/conv_time.py 10 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/call27.txt; cat gpurun_out/r5/call27.txt
