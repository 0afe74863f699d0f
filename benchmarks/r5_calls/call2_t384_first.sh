#!/bin/bash
# round 5, call 2: first run of gemm_nt_t384_kernel - parity, A/B against the 256 x 256 kernel, ablations, stamps
O=gpurun_out/r5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -k "t384 or large_tile or gemm_plain or epilogue" 2>&1 | tail -15 > $O/t384_pytest.txt
cat $O/t384_pytest.txt
( for rep in 1 2; do
    echo "== t384"; timeout 300 python benchmarks/gemm_time.py 5 plain --lab 2>&1 | grep TF
    echo "== 256x256 (DVD_GEMM_NO_T384)"; DVD_GEMM_NO_T384=1 timeout 300 python benchmarks/gemm_time.py 5 plain --lab 2>&1 | grep TF
  done
  for dbg in 1 2 3 4; do echo "== t384 ablation DVD_GEMM_T384_DBG=$dbg (1 no DMA, 2 no reads, 3 no barrier, 4 MFMA only)"; DVD_GEMM_T384_DBG=$dbg timeout 300 python benchmarks/gemm_time.py 5 plain --lab 2>&1 | grep TF; done
) > $O/t384_ab.txt 2>&1
cat $O/t384_ab.txt
( timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536; timeout 300 python benchmarks/gemm_t384_stamps.py 3072 1536 ) > $O/t384_stamps.txt 2>&1
cat $O/t384_stamps.txt
