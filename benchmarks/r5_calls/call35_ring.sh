#!/bin/bash
# round 5 call 35: the LDS-DMA ring kernel for f16 problems with few 128x128 tiles: tests (bits of the register-staged kernel), A/B, native point
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call35.txt
{
timeout 1500 python -m pytest tests/test_gpu_gemm.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -6
for d in 1 2 4; do python benchmarks/gemm_small_time.py $d 20 --lab 2>&1 | grep -v amdgpu.ids; done
for i in 1 2; do
echo "== single, product"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6
echo "== single, register-staged kernel"; DVD_GEMM_RING128=0 python benchmarks/native_profile.py 1 20 --lab 2>&1 | tail -6
done
} > $O 2>&1
cat $O
