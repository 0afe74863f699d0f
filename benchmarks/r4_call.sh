#!/bin/bash
mkdir -p gpurun_out/r4
export NOSTATS=1
export PROBE_OP=attn64
(timeout 300 python -m pytest tests/test_gpu_attention.py -q -x -k "h64m" 2>&1 | tail -2
timeout 300 python benchmarks/attn_ab.py 16 7 64 "glds64=" "h64m=DVD_ATTN_H64M" 2>&1 | tail -3
bash benchmarks/pmc_attn_ab.sh "glds64=" "h64m=DVD_ATTN_H64M" 2>&1 | tail -2) | tee gpurun_out/r4/c56_h64m.txt
