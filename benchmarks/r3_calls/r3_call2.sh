#!/bin/bash
# round-3 GPU call 2: LDS-tile grid_sample parity + timing, checkpoint-file test, dither drift record
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c2
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py -q -m gpu > gpurun_out/c2/t_ops.log 2>&1; echo "ops tests rc=$?"
python -m pytest tests/test_gpu_dropin.py -q -m gpu -k "checkpoint_files or without_prestage" > gpurun_out/c2/t_dropin.log 2>&1; echo "dropin tests rc=$?"
for B in 1 8; do
  python benchmarks/warp_time.py $B > gpurun_out/c2/warp_lds_B$B.txt 2>&1
  DVD_WARP_NOLDS=1 python benchmarks/warp_time.py $B --lab > gpurun_out/c2/warp_rows_B$B.txt 2>&1
done
cat gpurun_out/c2/warp_*.txt
python tests/tools/dither_drift.py > gpurun_out/c2/dither_drift.json 2> gpurun_out/c2/dither_drift.err; echo "drift rc=$?"
tail -3 gpurun_out/c2/t_ops.log; tail -3 gpurun_out/c2/t_dropin.log
cat gpurun_out/c2/dither_drift.json
