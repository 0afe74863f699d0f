#!/bin/bash
# round 5 call 17: the 128x128 GEMM with 4-deep register prefetch: tests, A/B at the native point's row counts, latency
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call17.txt
{
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q 2>&1 | tail -3
python benchmarks/gemm_small_time.py 1 20 --lab
python benchmarks/gemm_small_time.py 4 20 --lab
for i in 1 2; do
echo "== 4-deep"; python benchmarks/native_profile.py 1 20 --lab 2>&1 | tail -6
echo "== 2-deep"; DVD_GEMM_PD2=1 python benchmarks/native_profile.py 1 20 --lab 2>&1 | tail -6
done
echo "== 4-deep, 32"; python benchmarks/native_profile.py 32 5 --lab 2>&1 | tail -6
echo "== 2-deep, 32"; DVD_GEMM_PD2=1 python benchmarks/native_profile.py 32 5 --lab 2>&1 | tail -6
} > $O 2>&1
cat $O
