#!/bin/bash
# round 5 call 21: kernel profile of the native point at 32 documents per batch, at HEAD
cd /root/repo; mkdir -p gpurun_out/r5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof32 -o b32 -- python3 /root/repo/benchmarks/native_profile.py 32 5 > /root/repo/gpurun_out/r5/call21.txt 2>&1
cd /root/repo
cp $(find /tmp/prof32 -name "*kernel_stats.csv" | head -1) gpurun_out/r5/native_batch32_head_kernel_stats.csv
grep -A6 "documents per batch" gpurun_out/r5/call21.txt
