#!/bin/bash
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
