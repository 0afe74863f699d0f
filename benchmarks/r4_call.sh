#!/bin/bash
mkdir -p gpurun_out/r4
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], 'attn', d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])" | tee -a gpurun_out/r4/c73_boxes.txt
