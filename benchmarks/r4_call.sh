#!/bin/bash
mkdir -p gpurun_out/r4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r4/c62_gpu_tests_full.txt 2>&1; echo "rc=$?"
tail -14 gpurun_out/r4/c62_gpu_tests_full.txt
