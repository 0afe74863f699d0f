#!/usr/bin/env python3
"""Micro-benchmarks of single HIP ops (timed with events on torch's current stream, which is the
stream the ops are enqueued on).  Usage: python benchmarks/op_bench.py [unwarp] [gs] ..."""
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; _lab.use_lab()   # the DVD_* switches exist in the lab build only (make -C dvd_amd/csrc lab)
import sys
import json

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402

from dvd_amd import ops, synth  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench_unwarp():
    H, W, G = 3508, 2480, 288
    # document-like displacement: smooth (bicubic-upsampled 6x6 control points), +-0.05 normalised
    ctrl = torch.from_numpy(synth.uniform("b/flow", (1, 2, 6, 6), -0.05, 0.05, 1))
    flow = torch.nn.functional.interpolate(ctrl, size=(G, G), mode="bicubic", align_corners=True).contiguous().cuda()
    if os.environ.get("DVD_BENCH_RANDOM_FLOW"):
        flow = torch.from_numpy(synth.uniform("b/flow", (1, 2, G, G), -0.05, 0.05, 1)).cuda()
    import ctypes as C
    from dvd_amd import lib
    NSET = 3                       # rotate 3 buffer sets (> 256 MiB in total) so the Infinity Cache cannot hold them
    src8 = [torch.from_numpy(synth.synth_document(k, 8, 1, full_res=(H, W))["src_u8"]).cuda() for k in range(NSET)]
    srcf = [s8.permute(2, 0, 1)[None].float().contiguous() for s8 in src8]
    grid = [ops.unwarp_grid(flow, H, W) for _ in range(NSET)]
    outf = [torch.empty(H, W, 3, device="cuda") for _ in range(NSET)]
    out8 = [torch.empty(H, W, 3, dtype=torch.uint8, device="cuda") for _ in range(NSET)]
    outc = [torch.empty(1, 3, H, W, device="cuda") for _ in range(NSET)]
    st = lib.stream_ptr()
    px = H * W
    res = {}
    it = [0]

    def rot():
        it[0] = (it[0] + 1) % NSET
        return it[0]

    def f32():
        k = rot()
        lib.call("dvd_unwarp_f32", lib.ptr(flow), G, lib.ptr(srcf[k]), lib.ptr(outf[k]), H, W, C.c_float(0.987), st)

    def u8():
        k = rot()
        lib.call("dvd_unwarp_u8", lib.ptr(flow), G, lib.ptr(src8[k]), lib.ptr(out8[k]), H, W, C.c_float(0.987), st)

    def gs():
        k = rot()
        lib.call("dvd_grid_sample_bilinear_zeros_ac", lib.ptr(srcf[k]), lib.ptr(grid[k]), lib.ptr(outc[k]), 1, 3, H, W, H, W, 1, st)

    def gg():
        k = rot()
        lib.call("dvd_unwarp_grid", lib.ptr(flow), G, lib.ptr(grid[k]), H, W, C.c_float(0.987), st)

    for tag, extra in (("", {}), ("_scalar_fallback", {"DVD_WARP_SCALAR": "1"})):
        os.environ.update(extra)
        t = timeit(f32, iters=60)
        res["unwarp_f32_fused" + tag] = {"ms": t * 1e3, "GBps_algo(24B/px)": 24 * px / t / 1e9}
        t = timeit(u8, iters=60)
        res["unwarp_u8_fused" + tag] = {"ms": t * 1e3, "GBps_algo(6B/px)": 6 * px / t / 1e9}
        t = timeit(gs, iters=60)
        res["grid_sample_dropin" + tag] = {"ms": t * 1e3, "GBps_algo(32B/px)": 32 * px / t / 1e9}
        for k in extra:
            os.environ.pop(k)
    t = timeit(gg, iters=60)
    res["unwarp_grid"] = {"ms": t * 1e3, "GBps_algo(8B/px)": 8 * px / t / 1e9}
    a = torch.empty(px * 8, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    res["torch_copy_ref"] = {"ms": t * 1e3, "GBps": 2 * a.numel() * 4 / t / 1e9}
    return res


def bench_gemm():
    res = {}
    for (M, N, K, dt) in [(16 * 20736, 1536, 1536, torch.float16), (16 * 20736, 4608, 1536, torch.float16),
                          (16 * 20736, 2048, 1536, torch.float16), (4 * 16 * 20736 // 4, 1152, 384, torch.float16),
                          (2048, 1536, 1536, torch.float16), (8 * 20736, 384, 1536, torch.float32)]:
        a = torch.randn(M, K, device="cuda").to(dt)
        b = torch.randn(N, K, device="cuda").to(dt)
        out = torch.empty(M, N, dtype=torch.float16, device="cuda")
        t = timeit(lambda: ops.gemm_nt(a, b, out16=out), iters=5, warm=2)
        res[f"gemm_{'f16' if dt == torch.float16 else 'f32'}_{M}x{N}x{K}"] = {"ms": t * 1e3, "TFLOPs": 2.0 * M * N * K / t / 1e12}
    return res


def bench_attn():
    res = {}
    for (hd, B, T, scale) in [(256, 2, 20736, 0.0625), (64, 8, 20736, 0.125), (256, 2, 1024, 0.0625)]:
        C = 6 * hd
        qk = torch.randn(B, T, 2 * C, device="cuda").half()
        vt = torch.randn(B, C, T, device="cuda").half()
        out = torch.empty(B, T, C, dtype=torch.float16, device="cuda")
        t = timeit(lambda: ops.flash_attn(qk[:, :, :C], qk[:, :, C:], vt, out, 6, hd, scale), iters=3, warm=1)
        res[f"attn_hd{hd}_B{B}_T{T}"] = {"ms": t * 1e3, "TFLOPs": 4.0 * B * T * T * C / t / 1e12}
    return res


if __name__ == "__main__":
    which = sys.argv[1:] or ["unwarp"]
    out = {}
    if "unwarp" in which:
        out.update(bench_unwarp())
    if "attn" in which:
        out.update(bench_attn())
    if "gemm" in which:
        out.update(bench_gemm())
    print(json.dumps(out, indent=1))
