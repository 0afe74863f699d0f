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
*) echo "unknown call $1"; exit 2;;
esac
