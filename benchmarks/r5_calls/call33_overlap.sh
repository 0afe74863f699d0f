#!/bin/bash
# round 5 call 33: the engine's conv pyramid beside the nets (prestage.prepare_engine): tests, native point 1 / 32 with and without
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call33.txt
{
timeout 2400 python -m pytest tests/test_gpu_prestage.py tests/test_gpu_dropin.py tests/test_gpu_engine.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -5
for i in 1 2; do
echo "== single, separate"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6
echo "== single, overlapped"; python benchmarks/native_profile.py 1 20 --overlap 2>&1 | tail -7
done
echo "== 32, separate"; python benchmarks/native_profile.py 32 5 2>&1 | tail -6
echo "== 32, overlapped"; python benchmarks/native_profile.py 32 5 --overlap 2>&1 | tail -7
} > $O 2>&1
cat $O
