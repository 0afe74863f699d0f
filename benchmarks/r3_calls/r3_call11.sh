#!/bin/bash
# round-3 final check: smoke(), whole GPU suite, default bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c11
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c11/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/c11/smoke.log
( time python -m pytest tests -q -m gpu -x ) > gpurun_out/c11/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|^real" gpurun_out/c11/gpu_tests.log | tail -3
( time python bench.py ) > gpurun_out/c11/bench_default.json 2> gpurun_out/c11/bench_default.err; echo "bench rc=$?"
grep real gpurun_out/c11/bench_default.err
python -c "
import json; d=json.loads(open('gpurun_out/c11/bench_default.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['traffic'], d['roofline_unwarp']['achieved'], d['roofline_unwarp']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
