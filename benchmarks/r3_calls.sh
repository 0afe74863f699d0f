#!/bin/bash
# Round 3: the 16 gpurun command scripts of the round in the order they were run, collapsed into one record (VERDICT r4
# item 8; they were benchmarks/r3_calls/r3_call<N>.sh; each wrote under gpurun_out/, the cited summaries are profiles/r3_*;
# benchmarks/r3_final.sh / r3_final2.sh / r3_other_configs.sh produced the final-HEAD evidence).
# usage: bash benchmarks/r3_calls.sh <N>   - runs call N as it was issued (paths refer to the repository root).
case "$1" in
1)
# round-3 GPU call 1: new parity tests + dither on/off bench + kernel profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
python -m pytest tests/test_gpu_gemm.py -q -m gpu -x -k "dither" -s > gpurun_out/c1/t_dither.log 2>&1; echo "dither tests rc=$?"
python -m pytest tests/test_gpu_engine.py -q -m gpu -s > gpurun_out/c1/t_engine.log 2>&1; echo "engine tests rc=$?"
python -m pytest tests/test_gpu_dropin.py -q -m gpu -s -k "checkpoint_files or without_prestage" > gpurun_out/c1/t_dropin.log 2>&1; echo "dropin tests rc=$?"
python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/c1/bench_dither.json 2> gpurun_out/c1/bench_dither.err; echo "bench dither rc=$?"
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dither > gpurun_out/c1/bench_split.json 2> gpurun_out/c1/bench_split.err; echo "bench split rc=$?"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c1/prof" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/c1/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/c1/prof_bench.err"; echo "prof rc=$?"
cd "$GRAFT_REPO_ROOT"
python benchmarks/stats_summary.py "$(find gpurun_out/c1/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/c1/prof_bench.json "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (round 3, dithered weights)" > gpurun_out/c1/prof_summary.txt 2>&1
tail -3 gpurun_out/c1/t_dither.log; grep -E "passed|failed|error" gpurun_out/c1/t_engine.log | tail -3; tail -3 gpurun_out/c1/t_dropin.log
cat gpurun_out/c1/bench_dither.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dither', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
cat gpurun_out/c1/bench_split.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
head -20 gpurun_out/c1/prof_summary.txt
;;
2)
# round-3 GPU call 2: LDS-tile grid_sample parity + timing, checkpoint-file test, dither drift record
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c2
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py -q -m gpu > gpurun_out/c2/t_ops.log 2>&1; echo "ops tests rc=$?"
python -m pytest tests/test_gpu_dropin.py -q -m gpu -k "checkpoint_files or without_prestage" > gpurun_out/c2/t_dropin.log 2>&1; echo "dropin tests rc=$?"
for B in 1 8; do
  python benchmarks/warp_time.py $B > gpurun_out/c2/warp_lds_B$B.txt 2>&1
  DVD_WARP_NOLDS=1 python benchmarks/warp_time.py $B --lab > gpurun_out/c2/warp_rows_B$B.txt 2>&1
done
cat gpurun_out/c2/warp_*.txt
python tests/tools/dither_drift.py > gpurun_out/c2/dither_drift.json 2> gpurun_out/c2/dither_drift.err; echo "drift rc=$?"
tail -3 gpurun_out/c2/t_ops.log; tail -3 gpurun_out/c2/t_dropin.log
cat gpurun_out/c2/dither_drift.json
;;
3)
# round-3 GPU call 3: LDS-tile grid_sample variants, DDPM-250 on a large grid, weight families with dither
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c3
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py -q -m gpu -k "grid_sample or unwarp" > gpurun_out/c3/t_ops.log 2>&1; echo "ops tests rc=$?"
for B in 1 8; do
  python benchmarks/warp_time.py $B 2>/dev/null | head -1 > gpurun_out/c3/warp_lds16_B$B.txt
  DVD_WARP_LDSVAR=1 python benchmarks/warp_time.py $B --lab 2>/dev/null | head -1 > gpurun_out/c3/warp_lds32_B$B.txt
  DVD_WARP_NOLDS=1 python benchmarks/warp_time.py $B --lab 2>/dev/null | head -1 > gpurun_out/c3/warp_rows_B$B.txt
done
for f in gpurun_out/c3/warp_*.txt; do echo "$f: $(cat $f)"; done
( time python -m pytest tests/test_gpu_engine.py -q -m gpu -s -k "ddpm_250_steps_large" ) > gpurun_out/c3/t_ddpm.log 2>&1; echo "ddpm test rc=$?"
grep -E "ddpm 250|passed|failed|real" gpurun_out/c3/t_ddpm.log
( time python tests/tools/weight_sensitivity.py ) > gpurun_out/c3/weight_sensitivity.txt 2>&1; echo "families rc=$?"
cat gpurun_out/c3/weight_sensitivity.txt | grep -v amdgpu.ids
tail -3 gpurun_out/c3/t_ops.log
;;
4)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c4
rocprofv3 -L > gpurun_out/c4/counters.txt 2>&1 || rocprofv3-avail list > gpurun_out/c4/counters.txt 2>&1
bash benchmarks/pmc_warp.sh gpurun_out/c4/pmc_warp > gpurun_out/c4/pmc_warp_summary.txt 2>&1
cat gpurun_out/c4/pmc_warp_summary.txt
grep -c . gpurun_out/c4/counters.txt
;;
5)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c5
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py -q -m gpu -k "grid_sample or unwarp" > gpurun_out/c5/t_ops.log 2>&1; echo "ops tests rc=$?"; tail -2 gpurun_out/c5/t_ops.log
{
python benchmarks/warp_time.py 8 0.1 gs 2>/dev/null
for v in 1 2 3 4 5; do DVD_WARP_LDSVAR=$v python benchmarks/warp_time.py 8 0.1 gs --lab 2>/dev/null; done
for v in 3 5; do DVD_WARP_LDSVAR=$v python benchmarks/warp_time.py 8 0.0 gs --lab 2>/dev/null; done
DVD_WARP_NOLDS=1 python benchmarks/warp_time.py 8 0.1 gs --lab 2>/dev/null
} > gpurun_out/c5/warp_variants3.txt
cat gpurun_out/c5/warp_variants3.txt
;;
6)
# round-3 GPU call 6: default bench (with other_configs + 3-sample cpu_baseline), whole GPU suite, G=288 whole-loop parity
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c6
export TMPDIR=/tmp
( time python bench.py --steps 2 --warmup 1 ) > gpurun_out/c6/bench.json 2> gpurun_out/c6/bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/c6/bench.json | head -c 3000; echo
grep real gpurun_out/c6/bench.err
( time python -m pytest tests -q -m gpu -x ) > gpurun_out/c6/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -4 gpurun_out/c6/gpu_tests.log
( time python tests/tools/parity_g288.py 50 288 tame ) > gpurun_out/c6/parity_g288.json 2> gpurun_out/c6/parity_g288.err; echo "parity rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/c6/parity_g288.json").read().strip().splitlines()[0])
print({k: d[k] for k in ("last_x0_std", "saturated_pixels_last_x0", "oracle_seconds")})
for k, v in d["weights"].items():
    print(k, "final", v["final_coord_rmse"], "last x0", v["per_step_x0_rmse"][-1], "first", v["per_step_x0_rmse"][0])
PY
;;
7)
# round-3 GPU call 7: 16x16x32 GEMM (tests + timing A/B + bench), batched pre-stage (tests + timing)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c7
export TMPDIR=/tmp
python -m pytest tests/test_gpu_gemm.py -q -m gpu -x > gpurun_out/c7/t_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -3 gpurun_out/c7/t_gemm.log
python -m pytest tests/test_gpu_prestage.py -q -m gpu -x > gpurun_out/c7/t_prestage.log 2>&1; echo "prestage tests rc=$?"; tail -3 gpurun_out/c7/t_prestage.log
{
python benchmarks/gemm_time.py 7 plain 2>/dev/null
DVD_GEMM_M32=1 python benchmarks/gemm_time.py 7 plain --lab 2>/dev/null
python benchmarks/gemm_time.py 7 split 2>/dev/null
} > gpurun_out/c7/gemm_time.txt; cat gpurun_out/c7/gemm_time.txt
{
python benchmarks/prestage_time.py 64 1 2>/dev/null
python benchmarks/prestage_time.py 64 8 2>/dev/null
python benchmarks/prestage_time.py 64 16 2>/dev/null
} > gpurun_out/c7/prestage_time.txt; cat gpurun_out/c7/prestage_time.txt
python -m pytest tests/test_gpu_engine.py -q -m gpu -x -k "forward or golden or batched or long_loop" > gpurun_out/c7/t_engine.log 2>&1; echo "engine tests rc=$?"; tail -3 gpurun_out/c7/t_engine.log
python tests/tools/parity_g288.py 50 288 tame --engine-only gpurun_out/c7/g288_engine_traces.npz > gpurun_out/c7/g288_engine.log 2>&1; echo "traces rc=$?"; ls -la gpurun_out/c7/g288_engine_traces.npz
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/c7/bench_m16.json 2> gpurun_out/c7/bench_m16.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/c7/bench_m16.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_unwarp']['achieved'])"
;;
8)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c8
export TMPDIR=/tmp
{
python benchmarks/gemm_time.py 9 plain 2>&1 | grep -v amdgpu
DVD_GEMM_M32=1 python benchmarks/gemm_time.py 9 plain --lab 2>&1 | grep -v amdgpu
python benchmarks/gemm_time.py 9 plain 2>&1 | grep -v amdgpu
DVD_GEMM_M32=1 python benchmarks/gemm_time.py 9 plain --lab 2>&1 | grep -v amdgpu
} > gpurun_out/c8/gemm_time.txt; cat gpurun_out/c8/gemm_time.txt
;;
9)
# round-3 GPU call 9: whole GPU suite (durations), driver-like bench, rocprofv3 kernel stats, PMC round
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c9
export TMPDIR=/tmp
( time python -m pytest tests -q -m gpu --durations=25 ) > gpurun_out/c9/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|^real" gpurun_out/c9/gpu_tests.log | tail -3
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/c9/bench_driverlike.json 2> gpurun_out/c9/bench_driverlike.err; echo "bench rc=$?"
grep real gpurun_out/c9/bench_driverlike.err
python -c "
import json; d=json.loads(open('gpurun_out/c9/bench_driverlike.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_unwarp']['achieved'], d['cpu_baseline']['value'])"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c9/prof" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > "$GRAFT_REPO_ROOT/gpurun_out/c9/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/c9/prof_bench.err"; echo "prof rc=$?"
cd "$GRAFT_REPO_ROOT"
python benchmarks/stats_summary.py "$(find gpurun_out/c9/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/c9/prof_bench.json "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs (round 3 final kernels)" > gpurun_out/c9/prof_summary.txt 2>&1
head -14 gpurun_out/c9/prof_summary.txt | cut -c1-150
bash benchmarks/pmc_round.sh r3 > gpurun_out/c9/pmc_round.log 2>&1; echo "pmc rc=$?"
tail -12 gpurun_out/c9/pmc_round.log
;;
10)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c10
export TMPDIR=/tmp
{
for B in 16 64; do
python benchmarks/attn_time.py 64 $B 7 2>&1 | grep -v amdgpu
python benchmarks/attn_time.py 64 $B 7 --lib benchmarks/lab/alt/libdvd_hip_occ3.so 2>&1 | grep -v amdgpu
python benchmarks/attn_time.py 64 $B 7 2>&1 | grep -v amdgpu
python benchmarks/attn_time.py 64 $B 7 --lib benchmarks/lab/alt/libdvd_hip_occ3.so 2>&1 | grep -v amdgpu
done
} > gpurun_out/c10/attn64_occ.txt; cat gpurun_out/c10/attn64_occ.txt
python -m pytest tests/test_gpu_engine.py -q -m gpu -s -k "tame_family_loop_vs_reference" 2>&1 | grep -E "tame-family|passed|failed"
;;
11)
# round-3 final check: smoke(), whole GPU suite, default bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c11
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c11/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/c11/smoke.log
( time python -m pytest tests -q -m gpu -x ) > gpurun_out/c11/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|^real" gpurun_out/c11/gpu_tests.log | tail -3
( time python bench.py ) > gpurun_out/c11/bench_default.json 2> gpurun_out/c11/bench_default.err; echo "bench rc=$?"
grep real gpurun_out/c11/bench_default.err
python -c "
import json; d=json.loads(open('gpurun_out/c11/bench_default.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['traffic'], d['roofline_unwarp']['achieved'], d['roofline_unwarp']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
;;
12)
# round-3 GPU call 12: every per-step GEMM weight dithered (256 x 128 kernel on one weight tensor): tests, parity loops, profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c12
export TMPDIR=/tmp
python -m pytest tests/test_gpu_gemm.py -q -m gpu -x > gpurun_out/c12/t_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -2 gpurun_out/c12/t_gemm.log
python -m pytest tests/test_gpu_engine.py -q -m gpu -x -s -k "long_loop or ddpm_large or baseline_grid or graph_replay or batched" > gpurun_out/c12/t_engine.log 2>&1; echo "engine tests rc=$?"
grep -E "long loop rmse|split \(dither|plain f16|ddpm|G=288|passed|failed" gpurun_out/c12/t_engine.log | cut -c1-260
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c12/prof" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > "$GRAFT_REPO_ROOT/gpurun_out/c12/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/c12/prof_bench.err"; echo "prof rc=$?"
cd "$GRAFT_REPO_ROOT"
python benchmarks/stats_summary.py "$(find gpurun_out/c12/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/c12/prof_bench.json "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs (round 3: every per-step GEMM weight dithered)" > gpurun_out/c12/prof_summary.txt 2>&1
sed -n 3,14p gpurun_out/c12/prof_summary.txt | cut -c1-130
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/c12/bench.json 2> gpurun_out/c12/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/c12/bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
;;
13)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c13
export TMPDIR=/tmp
python benchmarks/ring_check.py 2>&1 | grep -v amdgpu
{
python benchmarks/gemm_time.py 9 plain 2>&1 | grep -v amdgpu
DVD_GEMM_RING=1 python benchmarks/gemm_time.py 9 plain --lab 2>&1 | grep -v amdgpu
python benchmarks/gemm_time.py 9 plain 2>&1 | grep -v amdgpu
DVD_GEMM_RING=1 python benchmarks/gemm_time.py 9 plain --lab 2>&1 | grep -v amdgpu
} > gpurun_out/c13/gemm_ring.txt; cat gpurun_out/c13/gemm_ring.txt
;;
14)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c14
export TMPDIR=/tmp
python -m pytest tests/test_gpu_tokens.py -q -m gpu -x 2>&1 | tail -2
python benchmarks/dwconv_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/c14/dwconv.txt
;;
15)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python benchmarks/fc1_time.py 2>&1 | grep -v amdgpu
python -m pytest tests/test_gpu_gemm.py -q -m gpu 2>&1 | tail -1
python -m pytest tests/test_gpu_engine.py -q -m gpu -s -k "forward or golden or long_loop" 2>&1 | grep -E "long loop rmse|golden forward|tame-family|passed|failed" | cut -c1-170
;;
16)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m pytest tests/test_gpu_engine.py -q -m gpu -s -k "forward or golden or batched or graph or long_loop or baseline_grid" 2>&1 | grep -E "long loop rmse|golden forward rmse|tame-family|G=288|passed|failed" | cut -c1-150
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
;;
*) echo "usage: $0 <1..16>"; exit 2 ;;
esac
