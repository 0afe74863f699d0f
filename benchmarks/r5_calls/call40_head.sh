#!/bin/bash
# round 5 call 40: whole GPU suite + the default bench run at HEAD (after the ring kernels)
cd /root/repo; mkdir -p gpurun_out/r5
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5/call40_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r5/call40_smoke.log
( time python -m pytest tests -q -m gpu --durations=6 ) > gpurun_out/r5/call40_gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|^real|FAILED" gpurun_out/r5/call40_gpu_tests.log | tail -5
( time python bench.py ) > gpurun_out/r5/call40_bench_default.json 2> gpurun_out/r5/call40_bench_default.err; echo "bench rc=$?"
grep real gpurun_out/r5/call40_bench_default.err
python -c "
import json; d=json.loads(open('gpurun_out/r5/call40_bench_default.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline_unwarp']['frac']); print({k: (v['value'], v['parity']['ok'] if v.get('parity') else None) for k, v in d['other_configs'].items()})"
