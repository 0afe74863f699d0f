import os, sys, torch, ctypes as C
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from dvd_amd import lib, ops, synth
H, W, G = 3508, 2480, 288
ctrl = torch.from_numpy(synth.uniform("b/flow", (1, 2, 6, 6), -0.05, 0.05, 1))
flow = torch.nn.functional.interpolate(ctrl, size=(G, G), mode="bicubic", align_corners=True).contiguous().cuda()
grid1 = ops.unwarp_grid(flow, H, W)
for n in (1, 2, 4, 8):
    NSET = 3 if n <= 2 else 2
    srcs = [torch.rand(n, 3, H, W, device="cuda") * 255 for _ in range(NSET)]
    grids = [grid1.repeat(n, 1, 1, 1).contiguous() for _ in range(NSET)]
    outs = [torch.empty(n, 3, H, W, device="cuda") for _ in range(NSET)]
    st = lib.stream_ptr()
    def f(k):
        lib.call("dvd_grid_sample_bilinear_zeros_ac", lib.ptr(srcs[k]), lib.ptr(grids[k]), lib.ptr(outs[k]), n, 3, H, W, H, W, 1, st)
    for i in range(3): f(i % NSET)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    it = 30
    for i in range(it): f(i % NSET)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / it
    print(f"grid_sample drop-in, n={n} images per launch: {ms:.3f} ms  {32 * H * W * n / ms / 1e9:.2f} TB/s ({32 * H * W * n / ms / 1e9 / 8 * 100:.0f} % of 8 TB/s)")
    del srcs, grids, outs
