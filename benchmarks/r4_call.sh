#!/bin/bash
mkdir -p gpurun_out/r4
timeout 300 python benchmarks/attn_ab.py 16 7 64 "add=" "pksum=DVD_ATTN_PKSUM" 2>&1 | tail -4 | tee gpurun_out/r4/c34_ab.txt
