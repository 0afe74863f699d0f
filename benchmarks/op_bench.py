#!/usr/bin/env python3
"""Micro-benchmarks of single HIP ops (timed with events on torch's current stream, which is the
stream the ops are enqueued on).  Usage: python benchmarks/op_bench.py [unwarp] [gs] ..."""
import os
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
    flow = torch.from_numpy(synth.uniform("b/flow", (1, 2, G, G), -0.05, 0.05, 1)).cuda()
    src8 = torch.from_numpy(synth.synth_document(0, 8, 1, full_res=(H, W))["src_u8"]).cuda()
    srcf = src8.permute(2, 0, 1)[None].float().contiguous()
    px = H * W
    res = {}
    t = timeit(lambda: ops.unwarp_f32(flow, srcf))
    res["unwarp_f32_fused"] = {"ms": t * 1e3, "GBps_algo(24B/px)": 24 * px / t / 1e9}
    t = timeit(lambda: ops.unwarp_u8(flow, src8))
    res["unwarp_u8_fused"] = {"ms": t * 1e3, "GBps_algo(6B/px)": 6 * px / t / 1e9}
    grid = ops.unwarp_grid(flow, H, W)
    t = timeit(lambda: ops.unwarp_grid(flow, H, W))
    res["unwarp_grid"] = {"ms": t * 1e3, "GBps_algo(8B/px)": 8 * px / t / 1e9}
    t = timeit(lambda: ops.grid_sample(srcf, grid))
    res["grid_sample_dropin"] = {"ms": t * 1e3, "GBps_algo(32B/px)": 32 * px / t / 1e9}
    a = torch.empty(px * 8, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    res["torch_copy_ref"] = {"ms": t * 1e3, "GBps": 2 * a.numel() * 4 / t / 1e9}
    return res


if __name__ == "__main__":
    which = sys.argv[1:] or ["unwarp"]
    out = {}
    if "unwarp" in which:
        out.update(bench_unwarp())
    print(json.dumps(out, indent=1))
