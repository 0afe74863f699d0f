#!/bin/bash
# round 5, call 7: the residual flavour (f32 in-place residual stream: the decoder's fc / conv2) - t384 vs the 256 x 256 kernel
O=gpurun_out/r5; mkdir -p $O
( for rep in 1 2; do
    echo "== t384 res"; timeout 300 python benchmarks/gemm_time.py 5 res --lab 2>&1 | grep TF
    echo "== 256x256 res (DVD_GEMM_NO_T384)"; DVD_GEMM_NO_T384=1 timeout 300 python benchmarks/gemm_time.py 5 res --lab 2>&1 | grep TF
  done
  echo "== t384 f32"; timeout 300 python benchmarks/gemm_time.py 5 f32 --lab 2>&1 | grep TF
  echo "== 256x256 f32"; DVD_GEMM_NO_T384=1 timeout 300 python benchmarks/gemm_time.py 5 f32 --lab 2>&1 | grep TF
) > $O/c7_res.txt 2>&1
cat $O/c7_res.txt
timeout 300 python benchmarks/gemm_t384_stamps.py 1536 1536 res 2>&1 | grep -v amdgpu.ids | tee $O/c7_res_stamps.txt
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -k "t384" 2>&1 | tail -3
# the reference's operating point, 32 documents per batch: stage split + kernel stats
timeout 600 python benchmarks/native_profile.py 32 5 2>&1 | grep -v amdgpu.ids | tee $O/c7_native32_stages.txt
cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r5/native32_prof" -o native32 -- python3 "$GRAFT_REPO_ROOT/benchmarks/native_profile.py" 32 3 > "$GRAFT_REPO_ROOT/gpurun_out/r5/c7_native32_prof.log" 2>&1; cd "$GRAFT_REPO_ROOT"
f=$(find gpurun_out/r5/native32_prof -name '*kernel_stats.csv' | head -1); echo "stats: $f"; head -25 "$f" | cut -c1-160
