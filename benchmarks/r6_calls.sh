#!/bin/bash
# Round 6: the gpurun command scripts of the round, in the order they were run (each writes under gpurun_out/r6/; the
# summaries cited by DESIGN.md are copied to profiles/r6_*).
# usage: bash benchmarks/r6_calls.sh <N>   - runs call N as it was issued (paths refer to the repository root).
O=gpurun_out/r6; mkdir -p $O
case "$1" in
1)   # the drop-in call shapes: register_model2([img, grid]), visualize_dewarping, _WrappedModel,
     # run_evaluation_docunet on a reference-shaped DataLoader
timeout 1200 python -m pytest tests/test_gpu_dropin.py -x -q ${K:+-k "$K"} > $O/c1_dropin.txt 2>&1; grep -v "^Starting\|^$\|Elapsed" $O/c1_dropin.txt | tail -60
;;
2)   # what bounds the residual epilogue of gemm_nt_t384_kernel: bytes in flight per CU, or the chip-wide burst?
( timeout 600 python benchmarks/gemm_t384_epi_probe.py 1536 2048; timeout 600 python benchmarks/gemm_t384_epi_probe.py 1536 1536 ) 2>&1 | grep -v Warning | tee $O/c2_epi_probe.txt
timeout 300 python benchmarks/dwconv_bench.py 2>&1 | tail -12 | tee $O/c2_dwconv_base.txt
;;
3)   # dwconv as a sliding window: bits vs the one-token kernel, then strip width x band height x nt stores
timeout 600 python -m pytest tests/test_gpu_tokens.py -x -q -k dwconv 2>&1 | tail -5 | tee $O/c3_dwconv_pytest.txt
timeout 600 python benchmarks/dwconv_bench.py 2>&1 | grep -v Warning | tee $O/c3_dwconv_bench.txt
;;
*) echo "unknown call $1"; exit 2;;
esac
