#!/bin/bash
# round 5, call 15: drop-in grid_sample (LDS-tile kernel): smaller LDS caps (more workgroups per CU) and direct global -> LDS box loads
O=gpurun_out/r5; mkdir -p $O
( for rep in 1 2; do
    for v in 0 6 7 8 9 10; do
      if [ $v = 0 ]; then echo "== product (cap 2048, register-staged)"; timeout 300 python benchmarks/warp_time.py 8 --lab 2>&1 | grep "grid_sample f32";
      else echo "== DVD_WARP_LDSVAR=$v (6: cap 1536, 7: cap 1024, 8: cap 2048 + LDS-DMA, 9: cap 1536 + DMA, 10: cap 1024 + DMA)"; DVD_WARP_LDSVAR=$v timeout 300 python benchmarks/warp_time.py 8 --lab 2>&1 | grep "grid_sample f32"; fi
    done
  done ) > $O/c15_warp.txt 2>&1
cat $O/c15_warp.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -2
