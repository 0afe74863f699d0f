#!/bin/bash
# round 5 call 18: the nets' narrow conv / GEMM kernels with three operand chunks in flight and back-to-back stores; the 128x128
# GEMM's fragment-read placement (lab variants); tests of everything they touch; native-point stages before/after
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call18.txt
{
timeout 1200 python -m pytest tests/test_gpu_prestage.py tests/test_gpu_gemm.py tests/test_gpu_ops.py -x -q 2>&1 | tail -3
python benchmarks/gemm_small_time.py 1 20 --lab
python benchmarks/gemm_small_time.py 4 20 --lab
for i in 1 2; do
echo "== single"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6
done
echo "== 32"; python benchmarks/native_profile.py 32 5 2>&1 | tail -6
} > $O 2>&1
cat $O
