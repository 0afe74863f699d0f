#!/bin/bash
# L2/fabric counters for the lab kernels (one pass per counter group)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/pmc_lab
rm -rf $out; mkdir -p $out
i=0
for c in ${PMC_GROUPS:-"FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"}; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $out/p$i -- benchmarks/lab/warp_lab > $out/p$i.log 2>&1 || tail -3 $out/p$i.log
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/pmc_lab/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):14.1f}")
PY
