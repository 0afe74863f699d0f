#!/usr/bin/env python3
"""Round 6: what bounds the residual epilogue of gemm_nt_t384_kernel - the CU's own bytes in flight, or the chip-wide burst?
Stamps (DVD_GEMM_T384_DBG=5) of the residual flavour with 256 / 128 / 64 / 32 persistent workgroups (DVD_GEMM_T384_NBLK: the work
per workgroup grows, the tile sequence of a workgroup does not change in kind) and with start-up staggers; per run: epilogue
cycles per tile (mean / p10 / p50 / p90), the mean number of workgroups that are inside their epilogue at the same time, and the
bytes per cycle a CU moves in it.    usage: python benchmarks/gemm_t384_epi_probe.py [N K] [M]"""
import os, subprocess, sys
if os.environ.get("EPI_CHILD") != "1":
    for nblk, stag in ((256, 0), (128, 0), (64, 0), (32, 0), (256, 8), (256, 16)):
        env = dict(os.environ, EPI_CHILD="1", DVD_GEMM_T384_DBG="5", DVD_GEMM_T384_NBLK=str(nblk), DVD_GEMM_T384_STAGGER=str(stag))
        subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; _lab.use_lab()
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import ctypes as C
import numpy as np
import torch
from dvd_amd import lib, ops
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1536, 2048)
M = int(sys.argv[3]) if len(sys.argv) > 3 else 331776
a = torch.randn(M, K, device="cuda").half(); b = torch.randn(N, K, device="cuda").half()
out = torch.zeros(M, N, dtype=torch.float32, device="cuda")
nt = (M // 384) * (N // 256)
st = torch.zeros(nt * 8 * 8, dtype=torch.int64, device="cuda")
lib.call("dvd_gemm_debug_stamps", C.c_void_p(st.data_ptr()))
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ops.gemm_nt(a, b, out32=out, res=out)
torch.cuda.synchronize()
ev0.record(); ops.gemm_nt(a, b, out32=out, res=out); ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1)
s = st.view(nt, 8, 8).cpu().numpy().astype(np.float64)
t0, t1, t2, t3 = s[:, :, 0], s[:, :, 1], s[:, :, 2], s[:, :, 3]
epi = (t3 - t2).mean(axis=1)                     # per tile (mean over its 8 waves)
loop = (t2 - t1).mean(axis=1)
# concurrency: at the midpoint of every tile's epilogue, how many tiles' epilogues are open
beg, end = t2.min(axis=1), t3.max(axis=1)
mid = 0.5 * (beg + end)
order = np.argsort(beg)
conc = np.array([np.sum((beg <= m) & (end >= m)) for m in mid[::max(1, nt // 512)]])
span = end.max() - t0.min()
print(f"NBLK={os.environ['DVD_GEMM_T384_NBLK']:>3} stagger={os.environ['DVD_GEMM_T384_STAGGER']:>2}: {ms:.3f} ms = "
      f"{2.0 * M * N * K / ms / 1e9:.0f} TF/s | epilogue per tile mean {epi.mean():.0f} p10 {np.percentile(epi, 10):.0f} p50 "
      f"{np.percentile(epi, 50):.0f} p90 {np.percentile(epi, 90):.0f} cycles = {786432 / epi.mean():.1f} B/clk per CU | K loop "
      f"{loop.mean():.0f} | epilogues open at once: mean {conc.mean():.0f} max {conc.max()} | kernel span {span:.0f} cycles "
      f"-> {span / ms / 1e3:.0f} MHz")
