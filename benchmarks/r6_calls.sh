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
4)   # hd-64 attention with the row sums on the matrix pipe: parity, soak, then A/B at the bench shape
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_tokens.py -x -q -k "h64x or H64X or dwconv" 2>&1 | tail -8 | tee $O/c4_h64l_pytest.txt
( timeout 600 python benchmarks/attn_ab.py 16 9 64 h64x_r4=DVD_ATTN_H64X_NOLM h64x=
  for a in 1 3; do echo "== ablation $a (1 no VALU, 3 MFMAs only)"; timeout 300 python benchmarks/attn_ab.py 16 3 64 h64x_r4=DVD_ATTN_H64X_NOLM,DVD_ATTN_H64X_ABL:$a h64x=DVD_ATTN_H64X_ABL:$a; done ) 2>&1 | grep -v Warning | tee $O/c4_h64l_ab.txt
;;
5)   # the product library with dwconv strips + h64x row sums on the matrix pipe: attention / tokens / engine suites, a short bench
timeout 2400 python -m pytest tests/test_gpu_attention.py tests/test_gpu_tokens.py tests/test_gpu_engine.py -x -q 2>&1 | tail -12 | tee $O/c5_pytest.txt
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs 2>$O/c5_bench.err | tail -1 > $O/c5_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6/c5_bench.json').read())
print('bench', d['value'], d['ms_per_step'], 'attn', d['roofline']['achieved'], d['roofline']['frac'])
print({k:v for k,v in d.items() if k in ('kernel_shares','gemm')})
PY
;;
6)   # PRICING the 16x16x32 MFMA shape for the 384 x 256 GEMM: two 16x16x32 MFMAs per 32x32x16 (garbage math, same FLOPs, LDS reads, DMA)
( for rep in 1 2; do for mode in plain f32; do
    echo "== 32x32x16 (product loop) $mode"; timeout 300 python benchmarks/gemm_time.py 5 $mode --lab 2>&1 | grep TF
    echo "== 16x16x32 pricing (DVD_GEMM_T384_DBG=6) $mode"; DVD_GEMM_T384_DBG=6 timeout 300 python benchmarks/gemm_time.py 5 $mode --lab 2>&1 | grep TF
  done; done ) | tee $O/c6_gemm_m16_pricing.txt
;;
7)   # the fused u8 tail as a column-block walk: bytes vs the row / scalar kernels, then the rates
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -5 | tee $O/c7_ops_pytest.txt
( echo "== product (band kernel)"; timeout 300 python benchmarks/warp_time.py 8 --lab 2>&1 | grep -v Warning
  echo "== DVD_WARP_U8_ROWS=1 (round 3's row kernel)"; DVD_WARP_U8_ROWS=1 timeout 300 python benchmarks/warp_time.py 8 --lab 2>&1 | grep -v Warning ) | tee $O/c7_warp_time.txt
;;
8)   # band height x unrolling of the u8 band kernel
( for ub in 2 4 8 16; do for un in 0 1; do
    if [ $un = 1 ]; then export DVD_WARP_U8_UNROLL=1; else unset DVD_WARP_U8_UNROLL; fi
    echo "== UB=$ub unroll=$un"; DVD_WARP_U8_UB=$ub timeout 300 python benchmarks/warp_time.py 8 --lab 2>&1 | grep "unwarp_u8"
  done; done; unset DVD_WARP_U8_UNROLL
  echo "== rows"; DVD_WARP_U8_ROWS=1 timeout 300 python benchmarks/warp_time.py 8 --lab 2>&1 | grep "unwarp_u8" ) | tee $O/c8_u8_band_variants.txt
;;
9)   # the 384 x 256 GEMM on v_mfma_f32_16x16x32_f16: parity, then A/B against round 5's loop (lab switch DVD_GEMM_T384_M32)
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -k "t384" 2>&1 | tail -15 | tee $O/c9_t384x_pytest.txt
( for rep in 1 2; do for mode in plain f32 res; do
    echo "== 16x16x32 $mode"; timeout 300 python benchmarks/gemm_time.py 5 $mode --lab 2>&1 | grep TF
    echo "== 32x32x16 (DVD_GEMM_T384_M32) $mode"; DVD_GEMM_T384_M32=1 timeout 300 python benchmarks/gemm_time.py 5 $mode --lab 2>&1 | grep TF
  done; done ) | tee $O/c9_t384x_ab.txt
( DVD_GEMM_T384_DBG=5 timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 f16; DVD_GEMM_T384_DBG=5 DVD_GEMM_T384_M32=1 timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 f16 ) 2>&1 | grep -v Warn | tee $O/c9_t384x_stamps.txt
;;
10)  # product library with the 16x16x32 GEMM loop: gemm / engine / dropin suites, A/B, a short bench
timeout 2400 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_engine.py tests/test_gpu_dropin.py tests/test_gpu_tokens.py tests/test_gpu_ops.py -x -q 2>&1 | tail -12 | tee $O/c10_pytest.txt
( for rep in 1 2; do for mode in plain f32 res; do
    echo "== 16x16x32 $mode"; timeout 300 python benchmarks/gemm_time.py 5 $mode --lab 2>&1 | grep TF
    echo "== 32x32x16 (DVD_GEMM_T384_M32) $mode"; DVD_GEMM_T384_M32=1 timeout 300 python benchmarks/gemm_time.py 5 $mode --lab 2>&1 | grep TF
  done; done ) > $O/c10_t384x_ab.txt; tail -16 $O/c10_t384x_ab.txt
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs 2>$O/c10_bench.err | tail -1 > $O/c10_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6/c10_bench.json').read())
print('bench', d['value'], d['ms_per_step'], 'attn', d['roofline']['achieved'], d['roofline']['frac'])
PY
;;
11)  # per-kernel split of one bench batch-step (rocprofv3 --kernel-trace --stats), mid-round
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r6/prof_mid" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > "$GRAFT_REPO_ROOT/gpurun_out/r6/prof_mid_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/r6/prof_mid_bench.err"; echo "prof rc=$?"
cd "$GRAFT_REPO_ROOT"
python benchmarks/stats_summary.py "$(find gpurun_out/r6/prof_mid -name '*kernel_stats.csv' | head -1)" gpurun_out/r6/prof_mid_bench.json "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs (round 6, mid-round)" > gpurun_out/r6/prof_mid_summary.txt 2>&1
sed -n 1,24p gpurun_out/r6/prof_mid_summary.txt | cut -c1-150; tail -12 gpurun_out/r6/prof_mid_summary.txt | cut -c1-200
find gpurun_out/r6/prof_mid -name '*kernel_stats.csv' -exec cp {} gpurun_out/r6/prof_mid_kernel_stats.csv \;
find gpurun_out/r6/prof_mid -name '*.csv' ! -name '*kernel_stats.csv' -delete 2>/dev/null
;;
12)  # stamps of the residual / f32 / f16 flavours, 16x16x32 vs 32x32x16 (lab), + quick parity
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -k "t384" 2>&1 | tail -3
( for mode in res f32 f16; do
    echo "== 16x16x32 $mode"; DVD_GEMM_T384_DBG=5 timeout 300 python benchmarks/gemm_t384_stamps.py 1536 2048 $mode 2>&1 | grep -v Warn | grep -v amdgpu
    echo "== 32x32x16 $mode"; DVD_GEMM_T384_DBG=5 DVD_GEMM_T384_M32=1 timeout 300 python benchmarks/gemm_t384_stamps.py 1536 2048 $mode 2>&1 | grep -v Warn | grep -v amdgpu
  done
  for mode in plain res; do echo "== 16x16x32 $mode"; timeout 300 python benchmarks/gemm_time.py 5 $mode --lab 2>&1 | grep TF; done ) | tee $O/c12_t384x_stamps.txt
;;
13)  # residual-window depth of the 16x16x32 kernel's residual epilogue (compile-time T384_RES_DEPTH; benchmarks/lab/alt/)
( echo "== depth 3 (product)"; timeout 300 python benchmarks/gemm_time.py 7 res --lab 2>&1 | grep TF
  for d in 4 5 6; do echo "== depth $d"; timeout 300 python benchmarks/gemm_time.py 7 res --lib benchmarks/lab/alt/libdvd_res_d$d.so 2>&1 | grep TF; done
  echo "== depth 3 (product)"; timeout 300 python benchmarks/gemm_time.py 7 res --lab 2>&1 | grep TF ) | tee $O/c13_res_depth.txt
;;
14)  # the whole GPU suite + smoke at the current commit
python -c "import __graft_entry__ as g; g.smoke()" > $O/c14_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/c14_smoke.log
( time python -m pytest tests -q -m gpu --durations=12 ) > $O/c14_gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|^real" $O/c14_gpu_tests.log | tail -3; grep -E "peaked|ddpm_g160" $O/c14_gpu_tests.log | head
;;
15)  # schedule variants of the 16x16x32 loop (generator switches T384X_PIECES / T384X_BAR; benchmarks/lab/alt/libdvd_t384x_v*.so)
timeout 300 python -m pytest tests/test_gpu_engine.py -q -k "ddpm_g160 or peaked" -s 2>&1 | grep -E "ddpm_g160|peaked|passed|failed" | cut -c1-600 | tee $O/c15_traces.txt
( for rep in 1 2; do
    echo "== product (pieces 3.1,3.3,3.5,3.7,4.1; barrier before block 3)"; timeout 300 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep TF
    for v in 1 2 3 4; do echo "== v$v"; timeout 300 python benchmarks/gemm_time.py 7 plain --lib benchmarks/lab/alt/libdvd_t384x_v$v.so 2>&1 | grep TF; done
  done ) | tee $O/c15_t384x_sched.txt
;;
final)  # round-6 evidence at HEAD: smoke(), whole GPU suite, PMC passes, the driver's bench invocation, rocprofv3 kernel stats
F=gpurun_out/final6; mkdir -p $F
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $F/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $F/smoke.log
( time python -m pytest tests -q -m gpu --durations=12 ) > $F/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|^real" $F/gpu_tests.log | tail -3
bash benchmarks/pmc_round.sh r6 > $F/pmc_round.log 2>&1; echo "pmc rc=$?"
cp gpurun_out/pmc_r6/r6_pmc_*.json gpurun_out/pmc_r6/r6_pmc_*.txt profiles/ 2>/dev/null
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $F/bench_driverlike.json 2> $F/bench_driverlike.err; echo "bench rc=$?"
grep real $F/bench_driverlike.err
python -c "
import json; d=json.loads(open('$F/bench_driverlike.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline_unwarp']['achieved'], d['roofline_unwarp']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores']); print({k: (v['value'], v['parity']['ok'] if v.get('parity') else None) for k, v in d['other_configs'].items()})"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$F/prof" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > "$GRAFT_REPO_ROOT/$F/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$F/prof_bench.err"; echo "prof rc=$?"
cd "$GRAFT_REPO_ROOT"
python benchmarks/stats_summary.py "$(find $F/prof -name '*kernel_stats.csv' | head -1)" $F/prof_bench.json "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs (round 6)" > $F/prof_summary.txt 2>&1
find $F/prof -name '*kernel_stats.csv' -exec cp {} $F/kernel_stats.csv \;
find $F/prof -name '*.csv' ! -name '*kernel_stats.csv' -delete 2>/dev/null
sed -n 3,12p $F/prof_summary.txt | cut -c1-130; tail -4 $F/prof_summary.txt
;;
18)  # the residual epilogue's loads alone (lab: DVD_GEMM_T384_NOSTORE) against loads + stores and stores alone (f32 flavour): stamps
( echo "== residual (loads + stores)"; DVD_GEMM_T384_DBG=5 timeout 300 python benchmarks/gemm_t384_stamps.py 1536 2048 res 2>&1 | grep -v Warn | grep -v amdgpu
  echo "== residual, stores never taken (loads alone)"; DVD_GEMM_T384_NOSTORE=1 DVD_GEMM_T384_DBG=5 timeout 300 python benchmarks/gemm_t384_stamps.py 1536 2048 res 2>&1 | grep -v Warn | grep -v amdgpu
  echo "== f32 flavour (stores alone)"; DVD_GEMM_T384_DBG=5 timeout 300 python benchmarks/gemm_t384_stamps.py 1536 2048 f32 2>&1 | grep -v Warn | grep -v amdgpu
  for v in "" DVD_GEMM_T384_NOSTORE; do echo "== wall, $v"; env ${v:+$v=1} timeout 300 python benchmarks/gemm_time.py 5 res --lab 2>&1 | grep TF; done ) | tee $O/c18_res_loads_alone.txt
;;
19)  # the residual epilogue in two phases (loads into the accumulators, then stores) against the interleaved form
     # (compile-time T384_RES_TWOPHASE=0: benchmarks/lab/alt/libdvd_res_interleaved.so - one epilogue per kernel instance)
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -k "t384" 2>&1 | tail -3
( for rep in 1 2; do
    echo "== two-phase (product rule, lab build)"; timeout 300 python benchmarks/gemm_time.py 7 res --lab 2>&1 | grep TF
    echo "== two-phase, window depth 2 (alt build)"; timeout 300 python benchmarks/gemm_time.py 7 res --lib benchmarks/lab/alt/libdvd_res_twophase_d2.so 2>&1 | grep TF
    echo "== interleaved (alt build)"; timeout 300 python benchmarks/gemm_time.py 7 res --lib benchmarks/lab/alt/libdvd_res_interleaved.so 2>&1 | grep TF
  done
  echo "== f32 flavour"; timeout 300 python benchmarks/gemm_time.py 7 f32 --lab 2>&1 | grep TF
  echo "== stamps two-phase"; DVD_GEMM_T384_DBG=5 timeout 300 python benchmarks/gemm_t384_stamps.py 1536 2048 res 2>&1 | grep -v Warn | grep -v amdgpu
) | tee $O/c19_res_twophase.txt
;;
*) echo "unknown call $1"; exit 2;;
esac
