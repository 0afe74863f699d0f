#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m pytest tests/test_gpu_engine.py -q -m gpu -s -k "forward or golden or batched or graph or long_loop or baseline_grid" 2>&1 | grep -E "long loop rmse|golden forward rmse|tame-family|G=288|passed|failed" | cut -c1-150
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
