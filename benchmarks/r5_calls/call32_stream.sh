#!/bin/bash
# round 5 call 32: side stream created once per device: prestage tests, native point 1 / 32
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call32.txt
{
timeout 1200 python -m pytest tests/test_gpu_prestage.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -4
for i in 1 2; do echo "== single"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6; done
echo "== 32"; python benchmarks/native_profile.py 32 5 2>&1 | tail -6
python benchmarks/latency_native.py --no-cpu 2>&1 | tail -3
} > $O 2>&1
cat $O
