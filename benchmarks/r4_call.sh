#!/bin/bash
mkdir -p gpurun_out/r4
(timeout 600 python -m pytest tests/test_gpu_tokens.py -q -x -k dwconv 2>&1 | tail -3
timeout 300 python benchmarks/dwconv_bench.py 2>&1 | grep -v amdgpu.ids) | tee gpurun_out/r4/c42_dwconv.txt
