#!/bin/bash
mkdir -p gpurun_out/r4
timeout 600 python benchmarks/attn_ab.py 16 5 256 "r64m=" "noeu=DVD_ATTN_R64M_ABL:5" "nodma=DVD_ATTN_R64M_ABL:6" "noread=DVD_ATTN_R64M_ABL:7" "mfmaonly=DVD_ATTN_R64M_ABL:4" "nobar=DVD_ATTN_R64M_ABL:2" 2>&1 | grep -v "max |" | tail -7 | tee gpurun_out/r4/c28_ab.txt
