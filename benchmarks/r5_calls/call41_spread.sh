#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5
python benchmarks/lab/big_spread_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/call41.txt; cat gpurun_out/r5/call41.txt
