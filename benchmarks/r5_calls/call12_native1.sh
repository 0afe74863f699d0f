#!/bin/bash
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
