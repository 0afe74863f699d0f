#!/bin/bash
# round 5 call 30: 33..64-channel convs on small maps as two 32-column halves: tests, native point 1 / 32
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call30.txt
{
timeout 2400 python -m pytest tests/test_gpu_tokens.py tests/test_gpu_prestage.py tests/test_gpu_dropin.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -5
for i in 1 2; do echo "== single"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6; done
echo "== 32"; python benchmarks/native_profile.py 32 5 2>&1 | tail -6
} > $O 2>&1
cat $O
