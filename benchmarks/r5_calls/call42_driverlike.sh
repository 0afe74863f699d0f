#!/bin/bash
# round 5 call 42: the driver's bench invocation at the last commit of the round
cd /root/repo; mkdir -p gpurun_out/r5
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5/call42_bench_driverlike.json 2> gpurun_out/r5/call42_bench_driverlike.err; echo "bench rc=$?"
grep real gpurun_out/r5/call42_bench_driverlike.err
python -c "
import json; d=json.loads(open('gpurun_out/r5/call42_bench_driverlike.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline_unwarp']['achieved'], d['roofline_unwarp']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores']); print({k: (v['value'], v['parity']['ok'] if v.get('parity') else None) for k, v in d['other_configs'].items()})"
