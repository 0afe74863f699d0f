set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -k big2 2>&1 | tail -15 > gpurun_out/r4/c11_gemm_tests.txt
cat gpurun_out/r4/c11_gemm_tests.txt
for i in 1 2; do
timeout 300 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep -v amdgpu.ids
DVD_GEMM_BIG2=1 timeout 300 python benchmarks/gemm_time.py 7 plain --lab 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r4/c11_gemm_time.txt
