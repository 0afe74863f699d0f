#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5
{ python benchmarks/gemm_ring128_stamps.py 2048 1536 1536; python benchmarks/gemm_ring128_stamps.py 2048 256 1536; python benchmarks/gemm_ring128_stamps.py 128 128 1536; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/call37.txt; cat gpurun_out/r5/call37.txt
