#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4/c67_gpu_tests_full.txt 2>&1; echo "rc=$?"
grep -E "passed|failed" gpurun_out/r4/c67_gpu_tests_full.txt | tail -3
timeout 900 python bench.py --steps 1 --warmup 1 --no-other-configs 2>gpurun_out/r4/c67_bench.err > gpurun_out/r4/c67_bench.json
python -c "
import json; d=json.loads(open('gpurun_out/r4/c67_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], [round(p['coord_rmse'],7) for p in d['cpu_baseline']['parity']])"
