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
# aggregates the round reports quote (bench point: 16 samples x 50 evaluations; FLOPs from bench.py:step_flops)
gem = sum(float(r["TotalDurationNs"]) for r in rows if "gemm_nt" in r["Name"]) / 1e9
a256 = sum(float(r["TotalDurationNs"]) for r in rows if "flash_attn" in r["Name"] and ("r64" in r["Name"] or "<256" in r["Name"])) / 1e9
a64 = sum(float(r["TotalDurationNs"]) for r in rows if "flash_attn" in r["Name"] and ("h64" in r["Name"] or "<64" in r["Name"])) / 1e9
steps = 1
if len(sys.argv) > 2:
    import json
    try:
        steps = int(json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]).get("steps", 1)) + int(
            json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]).get("warmup", 0))
    except Exception:
        pass
GEMM_TF, A256_TF, A64_TF = 16 * 50 * 4.272, 16 * 50 * 6 * 2.6418, 16 * 50 * 5.2836      # TFLOP per batch-step
if gem > 0:
    print(f"all GEMM kernels (gemm_nt*): {gem:.3f} s = {100 * gem / (tot / 1e9):.1f} % -> {GEMM_TF * steps / gem:.0f} TF/s algorithmic "
          f"({GEMM_TF * steps:.0f} TFLOP in {steps} batch-step(s))")
if a256 > 0:
    print(f"decoder attention (head_dim 256): {a256:.3f} s = {100 * a256 / (tot / 1e9):.1f} % -> {A256_TF * steps / a256:.0f} TF/s")
if a64 > 0:
    print(f"head_dim-64 attention: {a64:.3f} s = {100 * a64 / (tot / 1e9):.1f} % -> {A64_TF * steps / a64:.0f} TF/s")
