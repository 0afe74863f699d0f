#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c14
export TMPDIR=/tmp
python -m pytest tests/test_gpu_tokens.py -q -m gpu -x 2>&1 | tail -2
python benchmarks/dwconv_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/c14/dwconv.txt
