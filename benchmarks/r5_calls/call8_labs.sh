#!/bin/bash
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
