cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_attention.py -x -q -k "r64m" 2>&1 | tail -15
timeout 600 python benchmarks/attn_ab.py 16 7 256 r64p= r64m=DVD_ATTN_R64M 2>&1 | grep -v amdgpu.ids
