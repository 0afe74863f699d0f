#!/bin/bash
# the other BASELINE configurations at FULL size, once, with the final kernels (round 3)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/other
export TMPDIR=/tmp
python bench.py --docs 16 --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > gpurun_out/other/cfg4.json 2> gpurun_out/other/cfg4.err; echo "cfg4 rc=$?"
python bench.py --hyp 1 --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > gpurun_out/other/h1.json 2> gpurun_out/other/h1.err; echo "h1 rc=$?"
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs --no-dither > gpurun_out/other/split.json 2> gpurun_out/other/split.err; echo "split rc=$?"
python bench.py --docs 32 --sampler ddpm --ddim-steps 250 --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > gpurun_out/other/cfg3.json 2> gpurun_out/other/cfg3.err; echo "cfg3 rc=$?"
python - <<'PY'
import json
out = {}
for k in ("cfg4", "h1", "split", "cfg3"):
    d = json.loads(open(f"gpurun_out/other/{k}.json").read().strip().splitlines()[-1])
    out[k] = {"workload": d["config"]["workload"], "weights": d["config"]["weights"], "value": d["value"], "unit": d["unit"],
              "ms_per_step": d["ms_per_step"], "decoder_attention_tflops": d["roofline"]["achieved"]}
    print(k, out[k]["value"], out[k]["ms_per_step"], out[k]["decoder_attention_tflops"])
json.dump(out, open("gpurun_out/other/r3_other_configs.json", "w"), indent=1)
PY
