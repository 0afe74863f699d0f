#!/usr/bin/env python3
"""Time the full-resolution gathers (3508 x 2480, document-like flow) - drop-in grid_sample f32 (32 B/px), fused f32 tail
(24 B/px), fused u8 tail (6 B/px) - with B documents per launch.  usage: python benchmarks/warp_time.py [B=8]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; LIBSEL = _lab.which()   # --lab selects the lab build
import torch
from dvd_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
AMP = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1          # control-point range: 0.1 = the bench's +-0.05; 0 = identity
ONLY = sys.argv[3] if len(sys.argv) > 3 else ""                 # "gs": time the drop-in grid_sample only
H, W, G = 3508, 2480, 288
gen = torch.Generator(device="cuda").manual_seed(3)
ctrl = (torch.rand(B, 2, 6, 6, device="cuda", generator=gen) - 0.5) * AMP
flow = torch.nn.functional.interpolate(ctrl, size=(G, G), mode="bicubic", align_corners=True).contiguous()
src8 = torch.randint(0, 256, (B, H, W, 3), device="cuda", dtype=torch.uint8, generator=gen)
srcf = src8.permute(0, 3, 1, 2).float().contiguous()
grid = torch.cat([ops.unwarp_grid(flow[d:d + 1].contiguous(), H, W) for d in range(B)])
def t(f, n=7):
    for _ in range(2): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]
lib = LIBSEL
for name, f, bpp in (("grid_sample f32 (32 B/px)", lambda: ops.grid_sample(srcf, grid), 32),
                     ("unwarp_f32 fused (24 B/px)", lambda: ops.unwarp_f32_batch(flow, srcf), 24),
                     ("unwarp_u8 fused (6 B/px)", lambda: ops.unwarp_u8_batch(flow, src8), 6)):
    if ONLY == "gs" and not name.startswith("grid_sample"):
        continue
    ms = t(f)
    print(f"{name:28s} B={B}: {ms:.3f} ms  {bpp * H * W * B / ms / 1e6:.0f} GB/s   lib={lib} amp={AMP} var={os.environ.get('DVD_WARP_LDSVAR', '-')} wgs={os.environ.get('DVD_WARP_WGS', '-')} nolds={os.environ.get('DVD_WARP_NOLDS', '-')}")

if LIBSEL == "lab":
    # the streaming ceiling of the drop-in kernel's own access pattern (lab entry point; identity "gather")
    import ctypes as C
    from dvd_amd import lib
    from dvd_amd.lib import ptr, stream_ptr
    raw = lib.raw()
    raw.dvd_lab_stream_copy_planes.restype = C.c_int
    out = torch.empty_like(srcf)
    for tw in (32, 64, 128, 256):
        for nt in (0, 1):
            f = lambda: raw.dvd_lab_stream_copy_planes(ptr(srcf), ptr(grid), ptr(out), B, 3, H, W, nt, tw, stream_ptr())
            ms = t(f)
            print(f"{'stream copy ' + str(tw) + 'x' + str(1024 // tw) + (' nt' if nt else ''):28s} B={B}: {ms:.3f} ms  {32 * H * W * B / ms / 1e6:.0f} GB/s   "
                  f"(2 grid + 3 source planes read, 3 written, 16 B per lane, tiles in XCD bands, no gather)")
    f = lambda: out.copy_(srcf)
    ms = t(f)
    print(f"{'torch copy_ (D2D) of 3 planes':28s} B={B}: {ms:.3f} ms  {24 * H * W * B / ms / 1e6:.0f} GB/s  (24 B/px)")
