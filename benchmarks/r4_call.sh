cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_attention.py -x -q 2>&1 | tail -3
for i in 1 2 3; do python benchmarks/attn_time.py 64 16 9; python benchmarks/attn_time.py 64 16 9 --lib benchmarks/lab/alt/libdvd_hip_rowsum.so; done 2>&1 | grep -v amdgpu.ids
