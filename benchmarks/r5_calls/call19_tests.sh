#!/bin/bash
# round 5 call 19: tests of everything the narrow-kernel changes touch + kernel profile of the single-document native point
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call19.txt
{
timeout 1500 python -m pytest tests/test_gpu_prestage.py tests/test_gpu_gemm.py tests/test_gpu_ops.py tests/test_gpu_dropin.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -8
python benchmarks/gemm_small_time.py 1 20 --lab
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_single -o single -- python3 /root/repo/benchmarks/native_profile.py 1 10 2>&1 | tail -8
cd /root/repo
cp $(find /tmp/prof_single -name "*kernel_stats.csv" | head -1) gpurun_out/r5/native_single_kernel_stats.csv
} > $O 2>&1
cat $O
