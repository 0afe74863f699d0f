#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python benchmarks/fc1_time.py 2>&1 | grep -v amdgpu
python -m pytest tests/test_gpu_gemm.py -q -m gpu 2>&1 | tail -1
python -m pytest tests/test_gpu_engine.py -q -m gpu -s -k "forward or golden or long_loop" 2>&1 | grep -E "long loop rmse|golden forward|tame-family|passed|failed" | cut -c1-170
