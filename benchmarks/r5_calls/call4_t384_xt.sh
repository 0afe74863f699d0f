#!/bin/bash
# round 5, call 4: t384 with cross-tile prefetch + LDS-free epilogues; ATen-order warps (byte-exact u8 tail)
O=gpurun_out/r5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_ops.py -x -q 2>&1 | tail -12 > $O/c4_pytest.txt
cat $O/c4_pytest.txt
( for rep in 1 2; do
    echo "== t384"; timeout 300 python benchmarks/gemm_time.py 5 plain --lab 2>&1 | grep TF
    echo "== 256x256 (DVD_GEMM_NO_T384)"; DVD_GEMM_NO_T384=1 timeout 300 python benchmarks/gemm_time.py 5 plain --lab 2>&1 | grep TF
  done
  for dbg in 1 2 3 4; do echo "== t384 ablation DVD_GEMM_T384_DBG=$dbg (1 no DMA, 2 no reads, 3 no barrier, 4 MFMA only)"; DVD_GEMM_T384_DBG=$dbg timeout 300 python benchmarks/gemm_time.py 5 plain --lab 2>&1 | grep TF; done
) > $O/c4_ab.txt 2>&1
cat $O/c4_ab.txt
( timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536; timeout 300 python benchmarks/gemm_t384_stamps.py 3072 1536 ) > $O/c4_stamps.txt 2>&1
cat $O/c4_stamps.txt
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -k "baseline_grid or batched_documents or forward_stages or g96" 2>&1 | tail -5 | tee $O/c4_engine.txt
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/c4_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/c4_bench.json').read())
print('bench', d['value'], d['ms_per_step'], 'attn', d['roofline']['achieved'], d['roofline']['frac'], 'unwarp', d.get('roofline_unwarp',{}).get('frac'))
PY
