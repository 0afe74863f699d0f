#!/bin/bash
# round 5 call 39: the 128x256 ring kernel: tests (bits of the register-staged kernel), A/B at 1, 2, 4 documents' rows, native point
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call39.txt
{
timeout 1500 python -m pytest tests/test_gpu_gemm.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -6
for d in 1 2 4; do python benchmarks/gemm_small_time.py $d 20 --lab; done
for i in 1 2; do echo "== single, product"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6; done
} 2>&1 | grep -v amdgpu.ids > $O
cat $O
