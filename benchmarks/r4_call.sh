#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/r4/c26_gpu_tests.txt
timeout 900 python bench.py --steps 1 --warmup 1 2>gpurun_out/r4/c26_bench.err | tee gpurun_out/r4/c26_bench.json
