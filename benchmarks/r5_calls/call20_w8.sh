#!/bin/bash
# round 5 call 20: the 8-wave 128x128 GEMM: tests (bits of the 4-wave kernel), A/B at 1, 2, 4, 8 documents' rows, native-point stages
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call20.txt
{
timeout 1500 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_tokens.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -6
for d in 1 2 4 8; do python benchmarks/gemm_small_time.py $d 20 --lab 2>&1 | grep -v amdgpu.ids; done
for i in 1 2; do
echo "== single, product"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6
echo "== single, 4-wave"; DVD_GEMM_W8=0 python benchmarks/native_profile.py 1 20 --lab 2>&1 | tail -6
done
echo "== 32, product"; python benchmarks/native_profile.py 32 5 2>&1 | tail -6
} > $O 2>&1
cat $O
