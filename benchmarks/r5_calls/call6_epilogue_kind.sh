#!/bin/bash
# round 5, call 6: is the t384 epilogue bound by bytes or by store instructions?  (f16: 196 KB, f32: 393 KB, res: 393 + 393 KB per tile)
O=gpurun_out/r5; mkdir -p $O
( for m in f16 f32 res; do timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 $m 2>&1 | grep -v amdgpu.ids; done ) > $O/c6_epilogue_kind.txt 2>&1
cat $O/c6_epilogue_kind.txt
