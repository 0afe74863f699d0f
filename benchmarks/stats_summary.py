"""Turn a rocprofv3 *_kernel_stats.csv (+ the bench JSON line of the same run) into the text summary kept in profiles/.
usage: python benchmarks/stats_summary.py kernel_stats.csv bench.json "title" > profiles/<round>_summary.txt"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
print(sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py")
if len(sys.argv) > 2:
    print("bench line:", open(sys.argv[2]).read().strip())
print()
print(f"{'kernel':72s} {'calls':>6s} {'total s':>9s} {'avg ms':>10s} {'%':>6s}")
tot = 0.0
for r in rows:
    t = float(r["TotalDurationNs"])
    tot += t
    print(f"{r['Name'][:72]:72s} {r['Calls']:>6s} {t / 1e9:9.3f} {float(r['AverageNs']) / 1e6:10.3f} {float(r['Percentage']):6.1f}")
print(f"total kernel time {tot / 1e9:.3f} s")
