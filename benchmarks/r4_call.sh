#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r4/c52_gpu_tests_full.txt 2>&1; echo "rc=$?"
grep -E "passed|failed|error" gpurun_out/r4/c52_gpu_tests_full.txt | tail -5
