#!/bin/bash
# round 5, call 3: t384 with LDS-free epilogues (direct f32 / DPP-packed f16): parity, A/B, stamps, a first bench
O=gpurun_out/r5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q 2>&1 | tail -8 > $O/t384b_pytest.txt
cat $O/t384b_pytest.txt
( for rep in 1 2; do
    echo "== t384"; timeout 300 python benchmarks/gemm_time.py 5 plain --lab 2>&1 | grep TF
    echo "== 256x256 (DVD_GEMM_NO_T384)"; DVD_GEMM_NO_T384=1 timeout 300 python benchmarks/gemm_time.py 5 plain --lab 2>&1 | grep TF
  done ) > $O/t384b_ab.txt 2>&1
cat $O/t384b_ab.txt
( timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536; timeout 300 python benchmarks/gemm_t384_stamps.py 3072 1536 ) > $O/t384b_stamps.txt 2>&1
cat $O/t384b_stamps.txt
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -k "baseline_grid or batched_documents or forward_stages" 2>&1 | tail -5 | tee $O/t384b_engine.txt
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/t384b_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/t384b_bench.json').read())
print('bench', d['value'], d['ms_per_step'], 'attn', d['roofline']['achieved'], d['roofline']['frac'])
print({k:v for k,v in d.items() if k in ('kernel_shares','gemm')})
PY
