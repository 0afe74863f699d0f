#!/bin/bash
mkdir -p gpurun_out/r4
( time python bench.py ) > gpurun_out/r4/c58_bench_default.json 2> gpurun_out/r4/c58_bench_default.err; echo "rc=$?"
grep real gpurun_out/r4/c58_bench_default.err
python -c "
import json; d=json.loads(open('gpurun_out/r4/c58_bench_default.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['steps'], d['warmup'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline'].get('traffic'), d['cpu_baseline']['value'])"
