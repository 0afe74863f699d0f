#!/bin/bash
# round 5 call 29: smoke + the whole GPU suite at HEAD
cd /root/repo; mkdir -p gpurun_out/r5
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5/call29_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r5/call29_smoke.log
( time python -m pytest tests -q -m gpu --durations=8 ) > gpurun_out/r5/call29_gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|^real|FAILED|Error" gpurun_out/r5/call29_gpu_tests.log | tail -8
