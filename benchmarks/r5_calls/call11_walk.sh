#!/bin/bash
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
