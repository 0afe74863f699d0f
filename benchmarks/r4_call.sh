set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_dropin.py tests/test_gpu_gemm.py -x -q -m gpu --durations=12 --deselect "tests/test_gpu_engine.py::test_long_loop_vs_oracle_trace[ddim_g288_s50_tame]" -s 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r4/c10_engine_tests.txt
tail -45 gpurun_out/r4/c10_engine_tests.txt
