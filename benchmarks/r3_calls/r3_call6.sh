#!/bin/bash
# round-3 GPU call 6: default bench (with other_configs + 3-sample cpu_baseline), whole GPU suite, G=288 whole-loop parity
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c6
export TMPDIR=/tmp
( time python bench.py --steps 2 --warmup 1 ) > gpurun_out/c6/bench.json 2> gpurun_out/c6/bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/c6/bench.json | head -c 3000; echo
grep real gpurun_out/c6/bench.err
( time python -m pytest tests -q -m gpu -x ) > gpurun_out/c6/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -4 gpurun_out/c6/gpu_tests.log
( time python tests/tools/parity_g288.py 50 288 tame ) > gpurun_out/c6/parity_g288.json 2> gpurun_out/c6/parity_g288.err; echo "parity rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c6/parity_g288.json").read().strip().splitlines()[0])
print({k: d[k] for k in ("last_x0_std", "saturated_pixels_last_x0", "oracle_seconds")})
for k, v in d["weights"].items():
    print(k, "final", v["final_coord_rmse"], "last x0", v["per_step_x0_rmse"][-1], "first", v["per_step_x0_rmse"][0])
PY
