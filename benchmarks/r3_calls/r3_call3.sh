#!/bin/bash
# round-3 GPU call 3: LDS-tile grid_sample variants, DDPM-250 on a large grid, weight families with dither
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c3
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py -q -m gpu -k "grid_sample or unwarp" > gpurun_out/c3/t_ops.log 2>&1; echo "ops tests rc=$?"
for B in 1 8; do
  python benchmarks/warp_time.py $B 2>/dev/null | head -1 > gpurun_out/c3/warp_lds16_B$B.txt
  DVD_WARP_LDSVAR=1 python benchmarks/warp_time.py $B --lab 2>/dev/null | head -1 > gpurun_out/c3/warp_lds32_B$B.txt
  DVD_WARP_NOLDS=1 python benchmarks/warp_time.py $B --lab 2>/dev/null | head -1 > gpurun_out/c3/warp_rows_B$B.txt
done
for f in gpurun_out/c3/warp_*.txt; do echo "$f: $(cat $f)"; done
( time python -m pytest tests/test_gpu_engine.py -q -m gpu -s -k "ddpm_250_steps_large" ) > gpurun_out/c3/t_ddpm.log 2>&1; echo "ddpm test rc=$?"
grep -E "ddpm 250|passed|failed|real" gpurun_out/c3/t_ddpm.log
( time python tests/tools/weight_sensitivity.py ) > gpurun_out/c3/weight_sensitivity.txt 2>&1; echo "families rc=$?"
cat gpurun_out/c3/weight_sensitivity.txt | grep -v amdgpu.ids
tail -3 gpurun_out/c3/t_ops.log
