#!/usr/bin/env python3
"""Runs one launch-heavy op a few times so rocprofv3 --pmc can attribute counters to it.
usage: pmc_probe.py gemm|gemm_res|gemm_split|gemm_small|attn256|attn64|gridsample8|unwarp"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lab  # noqa: E402

LIBSEL = _lab.which()                   # --lab: the lab build (DVD_WARP_NOLDS=1 / DVD_WARP_LDSVAR=1 pick the gather variant)
from dvd_amd import ops, synth  # noqa: E402

which = sys.argv[1]
if which == "gemm":
    M, N, K = 331776, 1536, 1536
    a = torch.randn(M, K, device="cuda").half()
    b = torch.randn(N, K, device="cuda").half()
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    for _ in range(3):
        ops.gemm_nt(a, b, out16=out)
elif which == "gemm_res":                # the decoder's fc / conv2 form: f32 residual stream updated in place
    M, N, K = 331776, 1536, 1536
    a = torch.randn(M, K, device="cuda").half()
    b = (torch.randn(N, K, device="cuda") * 0.02).half()
    out = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    for _ in range(3):
        ops.gemm_nt(a, b, out32=out, res=out)
elif which == "gemm_split":
    M, N, K = 331776, 3072, 1536
    a = torch.randn(M, K, device="cuda").half()
    w = torch.randn(N, K, device="cuda") * 0.05
    hi = w.half(); lo = (w - hi.float()).half()
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    for _ in range(3):
        ops.gemm_nt(a, hi, out16=out, b_lo=lo, lo_scale=1.0)
elif which == "gemm_small":              # the sampler's GEMM at the reference's operating point: 2048 rows, (hi, lo) pair, 128 x 128 tiles
    M, N, K = 2048 * int(os.environ.get("PROBE_DOCS", "1")), 1536, 1536
    a = torch.randn(M, K, device="cuda").half()
    w = torch.randn(N, K, device="cuda") * 0.05
    hi = w.half(); lo = ((w - hi.float()) * 2048.0).half()
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    for _ in range(20):
        ops.gemm_nt(a, hi, out32=out, b_lo=lo, small_tiles=True)
elif which in ("attn256", "attn64"):  # attn256 at PROBE_B=16 is the bench launch shape (r64 kernel)
    hd = 256 if which == "attn256" else 64
    B, T = (int(os.environ.get("PROBE_B", "2")), 20736) if hd == 256 else (int(os.environ.get("PROBE_B", "8")), 20736)
    C = 6 * hd
    qk = torch.randn(B, T, 2 * C, device="cuda").half()
    vt = torch.randn(B, C, T, device="cuda").half()
    out = torch.empty(B, T, C, dtype=torch.float16, device="cuda")
    for _ in range(3):
        ops.flash_attn(qk[:, :, :C], qk[:, :, C:], vt, out, 6, hd, 1.0 / (hd ** 0.5))
elif which == "unwarp":
    H, W, G = 3508, 2480, 288
    ctrl = torch.from_numpy(synth.uniform("b/flow", (1, 2, 6, 6), -0.05, 0.05, 1))
    flow = torch.nn.functional.interpolate(ctrl, size=(G, G), mode="bicubic", align_corners=True).contiguous().cuda()
    src8 = torch.randint(0, 256, (H, W, 3), device="cuda", dtype=torch.uint8)
    srcf = src8.permute(2, 0, 1)[None].float().contiguous()
    grid = ops.unwarp_grid(flow, H, W)
    for _ in range(3):
        ops.unwarp_f32(flow, srcf)
        ops.unwarp_u8(flow, src8)
        ops.grid_sample(srcf, grid)
elif which == "gridsample8":             # the bench's roofline_unwarp leg: 8 documents per launch, drop-in f32 contract
    H, W, G, B = 3508, 2480, 288, 8
    gen = torch.Generator(device="cuda").manual_seed(3)
    ctrl = (torch.rand(B, 2, 6, 6, device="cuda", generator=gen) - 0.5) * 0.1
    flow = torch.nn.functional.interpolate(ctrl, size=(G, G), mode="bicubic", align_corners=True).contiguous()
    srcf = torch.rand(B, 3, H, W, device="cuda", generator=gen) * 255.0
    grid = torch.cat([ops.unwarp_grid(flow[d:d + 1].contiguous(), H, W) for d in range(B)])
    for _ in range(4):
        ops.grid_sample(srcf, grid)
elif which == "copy8":                   # the same byte count through torch's copy kernel, for the copy ceiling
    n = 8 * 32 * 3508 * 2480 // 8
    a = torch.rand(n, device="cuda")
    b = torch.empty_like(a)
    for _ in range(4):
        b.copy_(a)
torch.cuda.synchronize()
