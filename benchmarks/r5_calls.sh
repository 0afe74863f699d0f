#!/bin/bash
# Round 5: the gpurun command scripts of the round in the order they were run, collapsed into one record (they were
# benchmarks/r5_calls/call<N>_<what>.sh; each wrote under gpurun_out/r5/, the cited summaries are profiles/r5_*;
# benchmarks/r5_final.sh produced the evidence runs at HEAD).  Some calls were run more than once while a kernel changed
# (the profiles say which run they quote).
# usage: bash benchmarks/r5_calls.sh <N>[b]   - runs call N as it was issued (paths refer to the repository root).
case "$1" in
1)   # was call1_gemm_ablation.sh
# round 5, call 1: where does gemm_nt_big_kernel's time go?    -> profiles/r5_gemm_ablation.txt
O=gpurun_out/r5; mkdir -p $O
( ./benchmarks/lab/l2path_lab ) > $O/l2path_lab.txt 2>&1
( for dbg in 0 1 5 6 7 2 0; do echo "== DVD_GEMM_DEBUG=$dbg"; DVD_GEMM_DEBUG=$dbg timeout 300 python benchmarks/gemm_time.py 5 plain --lab 2>&1 | grep -v Warning; done ) > $O/gemm_ablation_wall.txt 2>&1
( DVD_GEMM_DEBUG=3 timeout 300 python benchmarks/gemm_stamps.py ) > $O/gemm_stamps.txt 2>&1
tail -50 $O/l2path_lab.txt $O/gemm_stamps.txt
;;
2)   # was call2_t384_first.sh
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
;;
3)   # was call3_t384_epilogues.sh
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
;;
4)   # was call4_t384_xt.sh
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
;;
5)   # was call5_stagger.sh
# round 5, call 5: start-up stagger of gemm_nt_t384_kernel (quantum x 1024 cycles x 0..15)
O=gpurun_out/r5; mkdir -p $O
( for q in 0 1 2 4 0 8 1 2; do echo "== DVD_GEMM_T384_STAGGER=$q"; DVD_GEMM_T384_STAGGER=$q timeout 300 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep TF; done ) > $O/c5_stagger.txt 2>&1
cat $O/c5_stagger.txt
( for q in 0 2; do echo "== stamps, stagger $q"; DVD_GEMM_T384_STAGGER=$q timeout 300 python benchmarks/gemm_t384_stamps.py 3072 1536; done ) > $O/c5_stagger_stamps.txt 2>&1
cat $O/c5_stagger_stamps.txt
;;
6)   # was call6_epilogue_kind.sh
# round 5, call 6: is the t384 epilogue bound by bytes or by store instructions?  (f16: 196 KB, f32: 393 KB, res: 393 + 393 KB per tile)
O=gpurun_out/r5; mkdir -p $O
( for m in f16 f32 res; do timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 $m 2>&1 | grep -v amdgpu.ids; done ) > $O/c6_epilogue_kind.txt 2>&1
cat $O/c6_epilogue_kind.txt
;;
7)   # was call7_res.sh
# round 5, call 7: the residual flavour (f32 in-place residual stream: the decoder's fc / conv2) - t384 vs the 256 x 256 kernel
O=gpurun_out/r5; mkdir -p $O
( for rep in 1 2; do
    echo "== t384 res"; timeout 300 python benchmarks/gemm_time.py 5 res --lab 2>&1 | grep TF
    echo "== 256x256 res (DVD_GEMM_NO_T384)"; DVD_GEMM_NO_T384=1 timeout 300 python benchmarks/gemm_time.py 5 res --lab 2>&1 | grep TF
  done
  echo "== t384 f32"; timeout 300 python benchmarks/gemm_time.py 5 f32 --lab 2>&1 | grep TF
  echo "== 256x256 f32"; DVD_GEMM_NO_T384=1 timeout 300 python benchmarks/gemm_time.py 5 f32 --lab 2>&1 | grep TF
) > $O/c7_res.txt 2>&1
cat $O/c7_res.txt
timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 res 2>&1 | grep -v amdgpu.ids | tee $O/c7_res_stamps.txt
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -k "t384" 2>&1 | tail -3
# the reference's operating point, 32 documents per batch: stage split + kernel stats
timeout 600 python benchmarks/native_profile.py 32 5 2>&1 | grep -v amdgpu.ids | tee $O/c7_native32_stages.txt
cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r5/native32_prof" -o native32 -- python3 "$GRAFT_REPO_ROOT/benchmarks/native_profile.py" 32 3 > "$GRAFT_REPO_ROOT/gpurun_out/r5/c7_native32_prof.log" 2>&1; cd "$GRAFT_REPO_ROOT"
f=$(find gpurun_out/r5/native32_prof -name '*kernel_stats.csv' | head -1); echo "stats: $f"; head -25 "$f" | cut -c1-160
;;
8)   # was call8_labs.sh
# round 5, call 8: exp-offload lab, decoder-attention XCD map experiment (item 7), t384 re-check
O=gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
./benchmarks/lab/exp_lab > $O/c8_exp_lab.txt 2>&1; cat $O/c8_exp_lab.txt
timeout 600 python benchmarks/attn_ab.py 16 7 256 product= spread=DVD_ATTN_XCDMAP:1 paired=DVD_ATTN_XCDMAP:2 2>&1 | grep -v amdgpu.ids > $O/c8_xcdmap_wall.txt; cat $O/c8_xcdmap_wall.txt
export PROBE_B=16
for m in 0 1 2; do
  DVD_ATTN_XCDMAP=$m timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/xcdmap_fetch_$m -- python3 benchmarks/pmc_probe.py attn256 --lab > $O/xcdmap_fetch_$m.log 2>&1
done
python3 - <<'PY' | tee gpurun_out/r5/c8_xcdmap_fetch.txt
import csv, glob
for m in (0, 1, 2):
    v = []
    for f in glob.glob(f"gpurun_out/r5/xcdmap_fetch_{m}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "flash_attn" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                v.append(float(r["Counter_Value"]))
    if v:
        print(f"DVD_ATTN_XCDMAP={m}: FETCH_SIZE {sum(v)/len(v):.0f} KiB per launch -> fabric reads {2*sum(v)/len(v)*1024/1e9:.2f} GB (x2: gfx950 FETCH_SIZE tallies 128-B requests at 64 B); {len(v)} launches")
PY
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -k "t384" 2>&1 | tail -2
;;
9)   # was call9_kloop_variants.sh
# round 5, call 9: K-loop schedule variants of gemm_nt_t384_kernel - where the five LDS-DMA pieces sit among the phase-2 MFMAs
# (generator switch T384_PIECES, alt builds) and static priority for waves 4-7 (DVD_GEMM_T384_PRIO)
O=gpurun_out/r5; mkdir -p $O
( for rep in 1 2; do
    echo "== pieces after MFMA 1,3,5,7,9 (product)"; timeout 300 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep TF
    for t in p0 p2 p3; do echo "== alt $t"; timeout 300 python benchmarks/gemm_time.py 7 plain --lib benchmarks/lab/alt/libdvd_t384_$t.so 2>&1 | grep TF; done
    echo "== product + s_setprio 1 for waves 4-7"; DVD_GEMM_T384_PRIO=1 timeout 300 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep TF
  done ) > $O/c9_kloop_variants.txt 2>&1
cat $O/c9_kloop_variants.txt
;;
10)   # was call10_nt_res.sh
# round 5, call 10: are streaming (nt) stores acknowledged sooner?  residual flavour of gemm_nt_t384_kernel, stamps + wall
O=gpurun_out/r5; mkdir -p $O
( for rep in 1 2; do
    echo "== res, plain stores"; timeout 300 python benchmarks/gemm_time.py 5 res --lab 2>&1 | grep TF
    echo "== res, nt stores"; DVD_GEMM_T384_NT=1 timeout 300 python benchmarks/gemm_time.py 5 res --lab 2>&1 | grep TF
  done
  echo "== stamps plain"; timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 res 2>&1 | grep -v amdgpu.ids
  echo "== stamps nt"; DVD_GEMM_T384_NT=1 timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 res 2>&1 | grep -v amdgpu.ids
) > $O/c10_nt_res.txt 2>&1
cat $O/c10_nt_res.txt
;;
11)   # was call11_walk.sh
# round 5, call 11: tile walk for the N = 1536 GEMMs (six N tiles): row-major vs two groups of three - wall, stamps, fabric traffic
O=gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
( for rep in 1 2 3; do
    for w in 0 1; do echo "== walk $w (plain f16)"; DVD_GEMM_T384_WALK=$w timeout 300 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep "fc \|c2 "; done
    for w in 0 1; do echo "== walk $w (res)"; DVD_GEMM_T384_WALK=$w timeout 300 python benchmarks/gemm_time.py 7 res --lab 2>&1 | grep "fc \|c2 "; done
  done ) > $O/c11_walk.txt 2>&1
cat $O/c11_walk.txt
for w in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    DVD_GEMM_T384_WALK=$w timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/walk${w}_$c -- python3 benchmarks/pmc_probe.py gemm --lab > $O/walk${w}_$c.log 2>&1
  done
done
python3 - <<'PY' | tee gpurun_out/r5/c11_walk_traffic.txt
import csv, glob
for w in (0, 1):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = []
        for f in glob.glob(f"gpurun_out/r5/walk{w}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_nt_t384" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    v.append(float(r["Counter_Value"]))
        tot[c] = sum(v) / max(len(v), 1)
    print(f"walk {w}: FETCH_SIZE {tot['FETCH_SIZE']:.0f} KiB, WRITE_SIZE {tot['WRITE_SIZE']:.0f} KiB -> (2 x FETCH + WRITE) = {(2*tot['FETCH_SIZE']+tot['WRITE_SIZE'])*1024/1e9:.2f} GB per launch (algorithmic 2.04 GB: A 1.02 + C 1.02)")
PY
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -k "t384" 2>&1 | tail -2
;;
12)   # was call12_native1.sh
# round 5, call 12: the reference's operating point, ONE document at a time: stage split + kernel time vs wall (is it launch-bound?)
O=gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python benchmarks/native_profile.py 1 15 2>&1 | grep -v amdgpu.ids | tee $O/c12_native1_stages.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/native1_prof -o native1 -- python3 benchmarks/native_profile.py 1 10 > $O/c12_native1_prof.log 2>&1
f=$(find $O/native1_prof -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY' | tee -a gpurun_out/r5/c12_native1_stages.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows if "dvd" in r["Name"] or "rocclr" in r["Name"])
calls = sum(int(r["Calls"]) for r in rows)
print(f"kernel time of 12 runs (2 warm-up + 10): {tot/1e6:.1f} ms = {tot/1e6/12:.2f} ms per document; {calls/12:.0f} launches per document")
for r in rows[:14]:
    print(f"  {r['Name'][:80]:80s} {int(r['Calls'])/12:7.1f} calls/doc {float(r['TotalDurationNs'])/1e6/12:7.3f} ms/doc")
PY
;;
13)   # was call13_phased.sh
# round 5, call 13: the phased residual epilogue of gemm_nt_t384_kernel against the interleaved one (lab switch) and the 256 x 256 kernel
O=gpurun_out/r5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -k "t384" 2>&1 | tail -2
( for rep in 1 2; do
    echo "== t384 res, phased"; timeout 300 python benchmarks/gemm_time.py 5 res --lab 2>&1 | grep TF
    echo "== t384 res, interleaved"; DVD_GEMM_T384_RES_INTERLEAVED=1 timeout 300 python benchmarks/gemm_time.py 5 res --lab 2>&1 | grep TF
    echo "== 256x256 res"; DVD_GEMM_NO_T384=1 timeout 300 python benchmarks/gemm_time.py 5 res --lab 2>&1 | grep TF
  done
  echo "== stamps phased"; timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 res 2>&1 | grep -v amdgpu.ids
  echo "== stamps interleaved"; DVD_GEMM_T384_RES_INTERLEAVED=1 timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 res 2>&1 | grep -v amdgpu.ids
  echo "== stamps f16"; timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 f16 2>&1 | grep -v amdgpu.ids
) > $O/c13_phased.txt 2>&1
cat $O/c13_phased.txt
;;
14)   # was call14_stagger_res.sh
# round 5, call 14: the residual epilogue is an HBM burst (every CU reads + writes 393 + 393 KB at the same moment: 200 MB per tile round);
# does a start-up stagger spread it?  (quantum x 1024 cycles x 0..15 per workgroup; a tile takes ~130 k cycles)
O=gpurun_out/r5; mkdir -p $O
( for q in 0 2 4 8 0 12 4 8; do echo "== res, DVD_GEMM_T384_STAGGER=$q"; DVD_GEMM_T384_STAGGER=$q timeout 300 python benchmarks/gemm_time.py 7 res --lab 2>&1 | grep TF; done
  for q in 0 8; do echo "== stamps res, stagger $q"; DVD_GEMM_T384_STAGGER=$q timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 res 2>&1 | grep -v amdgpu.ids; done
  for q in 0 8; do echo "== f16, DVD_GEMM_T384_STAGGER=$q"; DVD_GEMM_T384_STAGGER=$q timeout 300 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep TF; done
) > $O/c14_stagger_res.txt 2>&1
cat $O/c14_stagger_res.txt
;;
15)   # was call15_warp.sh
# round 5, call 15: drop-in grid_sample (LDS-tile kernel): smaller LDS caps (more workgroups per CU) and direct global -> LDS box loads
O=gpurun_out/r5; mkdir -p $O
( for rep in 1 2; do
    for v in 0 6 7 8 9 10; do
      if [ $v = 0 ]; then echo "== product (cap 2048, register-staged)"; timeout 300 python benchmarks/warp_time.py 8 --lab 2>&1 | grep "grid_sample f32";
      else echo "== DVD_WARP_LDSVAR=$v (6: cap 1536, 7: cap 1024, 8: cap 2048 + LDS-DMA, 9: cap 1536 + DMA, 10: cap 1024 + DMA)"; DVD_WARP_LDSVAR=$v timeout 300 python benchmarks/warp_time.py 8 --lab 2>&1 | grep "grid_sample f32"; fi
    done
  done ) > $O/c15_warp.txt 2>&1
cat $O/c15_warp.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -2
;;
16)   # was call16_bperm.sh
# round 5, call 16: f16 epilogue of gemm_nt_t384_kernel with the packed words sorted by ds_bpermute (contiguous lanes per row segment)
O=gpurun_out/r5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -k "t384 or large_tile" 2>&1 | tail -2
( timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 f16 2>&1 | grep -v amdgpu.ids
  timeout 300 python benchmarks/gemm_t384_stamps.py 3072 1536 f16 2>&1 | grep -v amdgpu.ids
  for rep in 1 2; do
    echo "== t384"; timeout 300 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep TF
    echo "== 256x256"; DVD_GEMM_NO_T384=1 timeout 300 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep TF
  done ) > $O/c16_bperm.txt 2>&1
cat $O/c16_bperm.txt
;;
17)   # was call17_pd4.sh
# round 5 call 17: the 128x128 GEMM with 4-deep register prefetch: tests, A/B at the native point's row counts, latency
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call17.txt
{
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q 2>&1 | tail -3
python benchmarks/gemm_small_time.py 1 20 --lab
python benchmarks/gemm_small_time.py 4 20 --lab
for i in 1 2; do
echo "== 4-deep"; python benchmarks/native_profile.py 1 20 --lab 2>&1 | tail -6
echo "== 2-deep"; DVD_GEMM_PD2=1 python benchmarks/native_profile.py 1 20 --lab 2>&1 | tail -6
done
echo "== 4-deep, 32"; python benchmarks/native_profile.py 32 5 --lab 2>&1 | tail -6
echo "== 2-deep, 32"; DVD_GEMM_PD2=1 python benchmarks/native_profile.py 32 5 --lab 2>&1 | tail -6
} > $O 2>&1
cat $O
;;
18)   # was call18_small.sh
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
;;
19)   # was call19_tests.sh
# round 5 call 19: tests of everything the narrow-kernel changes touch + kernel profile of the single-document native point
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call19.txt
{
timeout 1500 python -m pytest tests/test_gpu_prestage.py tests/test_gpu_gemm.py tests/test_gpu_ops.py tests/test_gpu_dropin.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -8
python benchmarks/gemm_small_time.py 1 20 --lab
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_single -o single -- python3 /root/repo/benchmarks/native_profile.py 1 10 2>&1 | tail -8
cd /root/repo
cp $(find /tmp/prof_single -name "*kernel_stats.csv" | head -1) gpurun_out/r5/native_single_kernel_stats.csv
} > $O 2>&1
cat $O
;;
20)   # was call20_w8.sh
# round 5 call 20: the 8-wave 128x128 GEMM: tests (bits of the 4-wave kernel), A/B at 1, 2, 4, 8 documents' rows, native-point stages
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call20.txt
{
timeout 1500 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_tokens.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -6
for d in 1 2 4 8; do python benchmarks/gemm_small_time.py $d 20 --lab 2>&1 | grep -v amdgpu.ids; done
for i in 1 2; do
echo "== single, product"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6
echo "== single, 4-wave"; DVD_GEMM_W8=0 python benchmarks/native_profile.py 1 20 --lab 2>&1 | tail -6
done
echo "== 32, product"; python benchmarks/native_profile.py 32 5 2>&1 | tail -6
} > $O 2>&1
cat $O
;;
21)   # was call21_prof32.sh
# round 5 call 21: kernel profile of the native point at 32 documents per batch, at HEAD
cd /root/repo; mkdir -p gpurun_out/r5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof32 -o b32 -- python3 /root/repo/benchmarks/native_profile.py 32 5 > /root/repo/gpurun_out/r5/call21.txt 2>&1
cd /root/repo
cp $(find /tmp/prof32 -name "*kernel_stats.csv" | head -1) gpurun_out/r5/native_batch32_head_kernel_stats.csv
grep -A6 "documents per batch" gpurun_out/r5/call21.txt
;;
22)   # was call22_embed.sh
# round 5 call 22: embed / K / V projections batched over document groups: engine + drop-in tests, native point at 1 and 32
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call22.txt
{
timeout 2400 python -m pytest tests/test_gpu_engine.py tests/test_gpu_dropin.py tests/test_gpu_tokens.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -6
echo "== single"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6
echo "== 32"; python benchmarks/native_profile.py 32 5 2>&1 | tail -6
echo "== 8"; python benchmarks/native_profile.py 8 8 2>&1 | tail -6
} > $O 2>&1
cat $O
;;
23)   # was call23_pmc_small.sh
# round 5 call 23: what bounds the 128x128 GEMM at 2048 rows - PMC passes (never combined with a trace), then durations
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r5/pmc_small
rm -rf $out; mkdir -p $out
for docs in 1 4; do
export PROBE_DOCS=$docs
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $out/mfma_$docs -- python3 benchmarks/pmc_probe.py gemm_small > $out/mfma_$docs.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $out/lds_$docs -- python3 benchmarks/pmc_probe.py gemm_small > $out/lds_$docs.log 2>&1
timeout 300 rocprofv3 --pmc TA_BUSY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/mem_$docs -- python3 benchmarks/pmc_probe.py gemm_small > $out/mem_$docs.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_$docs -- python3 benchmarks/pmc_probe.py gemm_small > $out/fetch_$docs.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$docs -- python3 benchmarks/pmc_probe.py gemm_small > $out/stats_$docs.log 2>&1
done
python3 - <<'P'
import csv, glob, os, collections
out = "gpurun_out/r5/pmc_small"
for docs in (1, 4):
    print(f"== gemm_small, {2048 * docs} rows x 1536 x 1536, (hi, lo)")
    for grp in ("mfma", "lds", "mem", "fetch"):
        acc = collections.defaultdict(list)
        for f in glob.glob(f"{out}/{grp}_{docs}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_nt" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"  {grp:5s} {k:28s} launches {len(v):3d}  mean {sum(v) / len(v):16.1f}")
    for f in glob.glob(f"{out}/stats_{docs}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_nt" in r["Name"]:
                print(f"  stats {r['Name'][:60]} calls {r['Calls']} avg {float(r['AverageNs']) / 1e3:.1f} us")
P
;;
24)   # was call24_implicit.sh
# round 5 call 24: wide convs as implicit GEMMs (nets + pyramid), batched output transposes: tests, native point at 1 / 32
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call24.txt
{
timeout 2400 python -m pytest tests/test_gpu_tokens.py tests/test_gpu_prestage.py tests/test_gpu_dropin.py tests/test_gpu_engine.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -6
echo "== single"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6
echo "== 32"; python benchmarks/native_profile.py 32 5 2>&1 | tail -6
} > $O 2>&1
cat $O
;;
26)   # was call26_occ.sh
# round 5 call 26: narrow conv kernels at capped occupancy (lab: DVD_CONV_LDS = dynamic LDS bytes per workgroup, unused by the kernel)
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call26.txt
{
for lds in 0 40000 65536 131072; do
echo "== DVD_CONV_LDS=$lds (0: 3 workgroups per CU by registers; 40000: 4 -> still 3; 65536: 2; 131072: 1), 32 documents"
DVD_CONV_LDS=$lds python benchmarks/native_profile.py 32 4 --lab 2>&1 | grep "prestage\|prepare_docs\|documents per batch"
done
for lds in 0 65536 131072; do
echo "== DVD_CONV_LDS=$lds, 1 document"
DVD_CONV_LDS=$lds python benchmarks/native_profile.py 1 12 --lab 2>&1 | grep "prestage\|prepare_docs\|documents per batch"
done
} > $O 2>&1
cat $O
;;
27)   # was call27_conv.sh
cd /root/repo; mkdir -p gpurun_out/r5
python benchmarks/conv_time.py 10 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/call27.txt; cat gpurun_out/r5/call27.txt
;;
28)   # was call28_smalllin.sh
# round 5 call 28: small_linear over sample groups in parallel: engine / tokens tests, native point at 32 and 1
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call28.txt
{
timeout 2400 python -m pytest tests/test_gpu_tokens.py tests/test_gpu_engine.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -5
echo "== 32"; python benchmarks/native_profile.py 32 5 2>&1 | tail -6
echo "== single"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6
} > $O 2>&1
cat $O
;;
29)   # was call29_fullgpu.sh
# round 5 call 29: smoke + the whole GPU suite at HEAD
cd /root/repo; mkdir -p gpurun_out/r5
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5/call29_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r5/call29_smoke.log
( time python -m pytest tests -q -m gpu --durations=8 ) > gpurun_out/r5/call29_gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|^real|FAILED|Error" gpurun_out/r5/call29_gpu_tests.log | tail -8
;;
30)   # was call30_split.sh
# round 5 call 30: 33..64-channel convs on small maps as two 32-column halves: tests, native point 1 / 32
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call30.txt
{
timeout 2400 python -m pytest tests/test_gpu_tokens.py tests/test_gpu_prestage.py tests/test_gpu_dropin.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -5
for i in 1 2; do echo "== single"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6; done
echo "== 32"; python benchmarks/native_profile.py 32 5 2>&1 | tail -6
} > $O 2>&1
cat $O
;;
31)   # was call31_trace1.sh
# round 5 call 31: per-launch durations of one single-document pass at the native point (which launches are the long ones)
cd /root/repo; mkdir -p gpurun_out/r5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace1 -o t1 -- python3 /root/repo/benchmarks/native_profile.py 1 3 > /root/repo/gpurun_out/r5/call31.txt 2>&1
cd /root/repo
python3 - <<'P'
import csv, glob
f = glob.glob('/tmp/trace1/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last pass: take the last 1100 launches
last = rows[-1100:]
# find the start of the last run: ingest_resize kernels mark the beginning
idx = max(i for i, r in enumerate(last) if 'ingest_resize' in r['Kernel_Name'])
run = last[idx - 1:]
t0 = int(run[0]['Start_Timestamp'])
out = open('gpurun_out/r5/native_single_trace.txt', 'w')
tot = 0
for r in run:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    out.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {d:8.1f} us  grid {r['Grid_Size_X']:>8s}x{r['Grid_Size_Y']:>3s}  {r['Kernel_Name'][:90]}\n")
out.write(f"launches {len(run)}, kernel time {tot / 1e3:.2f} ms, span {(int(run[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms\n")
out.close()
print(open('gpurun_out/r5/native_single_trace.txt').read()[-300:])
P
;;
32)   # was call32_stream.sh
# round 5 call 32: side stream created once per device: prestage tests, native point 1 / 32
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call32.txt
{
timeout 1200 python -m pytest tests/test_gpu_prestage.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -4
for i in 1 2; do echo "== single"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6; done
echo "== 32"; python benchmarks/native_profile.py 32 5 2>&1 | tail -6
python benchmarks/latency_native.py --no-cpu 2>&1 | tail -3
} > $O 2>&1
cat $O
;;
33)   # was call33_overlap.sh
# round 5 call 33: the engine's conv pyramid beside the nets (prestage.prepare_engine): tests, native point 1 / 32 with and without
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call33.txt
{
timeout 2400 python -m pytest tests/test_gpu_prestage.py tests/test_gpu_dropin.py tests/test_gpu_engine.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -5
for i in 1 2; do
echo "== single, separate"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6
echo "== single, overlapped"; python benchmarks/native_profile.py 1 20 --overlap 2>&1 | tail -7
done
echo "== 32, separate"; python benchmarks/native_profile.py 32 5 2>&1 | tail -6
echo "== 32, overlapped"; python benchmarks/native_profile.py 32 5 --overlap 2>&1 | tail -7
} > $O 2>&1
cat $O
;;
34)   # was call34_trace.sh
# round 5 call 34: the two G = 288 oracle traces of round 5 against the engine
cd /root/repo; mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_gpu_engine.py -q -s -k "ddim_g288_s50_plain or ddpm_g288_s10_tame" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -12 > gpurun_out/r5/call34.txt
cat gpurun_out/r5/call34.txt
;;
35)   # was call35_ring.sh
# round 5 call 35: the LDS-DMA ring kernel for f16 problems with few 128x128 tiles: tests (bits of the register-staged kernel), A/B, native point
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call35.txt
{
timeout 1500 python -m pytest tests/test_gpu_gemm.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -6
for d in 1 2 4; do python benchmarks/gemm_small_time.py $d 20 --lab 2>&1 | grep -v amdgpu.ids; done
for i in 1 2; do
echo "== single, product"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6
echo "== single, register-staged kernel"; DVD_GEMM_RING128=0 python benchmarks/native_profile.py 1 20 --lab 2>&1 | tail -6
done
} > $O 2>&1
cat $O
;;
36)   # was call36_l2warm.sh
cd /root/repo; mkdir -p gpurun_out/r5
python benchmarks/gemm_l2warm.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/call36.txt; cat gpurun_out/r5/call36.txt
;;
37)   # was call37_stamps.sh
cd /root/repo; mkdir -p gpurun_out/r5
{ python benchmarks/gemm_ring128_stamps.py 2048 1536 1536; python benchmarks/gemm_ring128_stamps.py 2048 256 1536; python benchmarks/gemm_ring128_stamps.py 128 128 1536; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/call37.txt; cat gpurun_out/r5/call37.txt
;;
38)   # was call38_spread.sh
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call38.txt
{
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -k "ring128 or eight_wave or small_family" 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -3
python benchmarks/gemm_ring128_stamps.py 2048 1536 1536; python benchmarks/gemm_ring128_stamps.py 128 128 1536
for d in 1 2; do python benchmarks/gemm_small_time.py $d 20 --lab; done
echo "== single, product"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6
} 2>&1 | grep -v amdgpu.ids > $O
cat $O
;;
39)   # was call39_ring256.sh
# round 5 call 39: the 128x256 ring kernel: tests (bits of the register-staged kernel), A/B at 1, 2, 4 documents' rows, native point
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call39.txt
{
timeout 1500 python -m pytest tests/test_gpu_gemm.py -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP" | tail -6
for d in 1 2 4; do python benchmarks/gemm_small_time.py $d 20 --lab; done
for i in 1 2; do echo "== single, product"; python benchmarks/native_profile.py 1 20 2>&1 | tail -6; done
} 2>&1 | grep -v amdgpu.ids > $O
cat $O
;;
40)   # was call40_head.sh
# round 5 call 40: whole GPU suite + the default bench run at HEAD (after the ring kernels)
cd /root/repo; mkdir -p gpurun_out/r5
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5/call40_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r5/call40_smoke.log
( time python -m pytest tests -q -m gpu --durations=6 ) > gpurun_out/r5/call40_gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|^real|FAILED" gpurun_out/r5/call40_gpu_tests.log | tail -5
( time python bench.py ) > gpurun_out/r5/call40_bench_default.json 2> gpurun_out/r5/call40_bench_default.err; echo "bench rc=$?"
grep real gpurun_out/r5/call40_bench_default.err
python -c "
import json; d=json.loads(open('gpurun_out/r5/call40_bench_default.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline_unwarp']['frac']); print({k: (v['value'], v['parity']['ok'] if v.get('parity') else None) for k, v in d['other_configs'].items()})"
;;
41)   # was call41_spread.sh
cd /root/repo; mkdir -p gpurun_out/r5
python benchmarks/lab/big_spread_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/call41.txt; cat gpurun_out/r5/call41.txt
;;
42)   # was call42_driverlike.sh
# round 5 call 42: the driver's bench invocation at the last commit of the round
cd /root/repo; mkdir -p gpurun_out/r5
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5/call42_bench_driverlike.json 2> gpurun_out/r5/call42_bench_driverlike.err; echo "bench rc=$?"
grep real gpurun_out/r5/call42_bench_driverlike.err
python -c "
import json; d=json.loads(open('gpurun_out/r5/call42_bench_driverlike.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline_unwarp']['achieved'], d['roofline_unwarp']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores']); print({k: (v['value'], v['parity']['ok'] if v.get('parity') else None) for k, v in d['other_configs'].items()})"
;;
*) echo "usage: bash benchmarks/r5_calls.sh <N>"; exit 2 ;;
esac
