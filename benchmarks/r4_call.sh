#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_attention.py -q 2>&1 | tail -3 | tee gpurun_out/r4/c61_tests.txt
