set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_abi.py -x -q 2>&1 | tail -15 > gpurun_out/r4/c9_attn_tests.txt
timeout 600 python benchmarks/attn_ab.py 16 7 256 > gpurun_out/r4/c9_ab.txt 2>&1
cat gpurun_out/r4/c9_attn_tests.txt gpurun_out/r4/c9_ab.txt
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/r4/c9_bench.json 2> gpurun_out/r4/c9_bench.err
tail -c 3000 gpurun_out/r4/c9_bench.json
