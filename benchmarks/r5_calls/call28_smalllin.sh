#!/bin/bash
# round 5 call 28: small_linear over sample groups in parallel: engine / tokens tests, native point at 32 and 1
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call28.txt
{
timeout 2400 python -m pytest tests/test_gpu_tokens.py tests/test_gpu_engine.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -5
echo "== 32"; python benchmarks/native_profile.py 32 5 2>&1 | tail -6
echo "== single"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6
} > $O 2>&1
cat $O
