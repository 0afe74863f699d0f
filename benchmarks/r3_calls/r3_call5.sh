#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c5
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py -q -m gpu -k "grid_sample or unwarp" > gpurun_out/c5/t_ops.log 2>&1; echo "ops tests rc=$?"; tail -2 gpurun_out/c5/t_ops.log
{
python benchmarks/warp_time.py 8 0.1 gs 2>/dev/null
for v in 1 2 3 4 5; do DVD_WARP_LDSVAR=$v python benchmarks/warp_time.py 8 0.1 gs --lab 2>/dev/null; done
for v in 3 5; do DVD_WARP_LDSVAR=$v python benchmarks/warp_time.py 8 0.0 gs --lab 2>/dev/null; done
DVD_WARP_NOLDS=1 python benchmarks/warp_time.py 8 0.1 gs --lab 2>/dev/null
} > gpurun_out/c5/warp_variants3.txt
cat gpurun_out/c5/warp_variants3.txt
