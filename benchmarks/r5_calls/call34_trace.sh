#!/bin/bash
# round 5 call 34: the two G = 288 oracle traces of round 5 against the engine
cd /root/repo; mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_gpu_engine.py -q -s -k "ddim_g288_s50_plain or ddpm_g288_s10_tame" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -12 > gpurun_out/r5/call34.txt
cat gpurun_out/r5/call34.txt
