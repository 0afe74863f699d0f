#!/bin/bash
# round 5 call 24: wide convs as implicit GEMMs (nets + pyramid), batched output transposes: tests, native point at 1 / 32
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call24.txt
{
timeout 2400 python -m pytest tests/test_gpu_tokens.py tests/test_gpu_prestage.py tests/test_gpu_dropin.py tests/test_gpu_engine.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -6
echo "== single"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6
echo "== 32"; python benchmarks/native_profile.py 32 5 2>&1 | tail -6
} > $O 2>&1
cat $O
