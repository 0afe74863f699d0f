#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call38.txt
{
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -k "ring128 or eight_wave or small_family" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -3
python benchmarks/gemm_ring128_stamps.py 2048 1536 1536; python benchmarks/gemm_ring128_stamps.py 128 128 1536
for d in 1 2; do python benchmarks/gemm_small_time.py $d 20 --lab; done
echo "== single, product"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6
} 2>&1 | grep -v amdgpu.ids > $O
cat $O
