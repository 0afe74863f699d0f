#!/usr/bin/env python3
"""Time the decoder's GEMMs at the bench's shapes (M = 331 776 rows): plain f16 weights (what the dithered engine runs;
product = the 32x32x16 kernel, `--lab` with DVD_GEMM_M16=1 = the rejected 16x16x32 variant) and, with `split`, the (hi, lo) pairs.
usage: python benchmarks/gemm_time.py [reps=5] [plain|split|res|f32]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; LIBSEL = _lab.which()   # --lab selects the lab build
import torch
from dvd_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
MODE = sys.argv[2] if len(sys.argv) > 2 else "plain"
EXE = 4.0 if MODE == "split" else 2.0
M32 = os.environ.get("DVD_GEMM_M16", "-") + ("/big2" if os.environ.get("DVD_GEMM_BIG2") else "")
M = 331776
for name, N, K in (("qk  N=3072 K=1536", 3072, 1536), ("c1  N=2048 K=1536", 2048, 1536), ("fc  N=1536 K=1536", 1536, 1536),
                   ("c2  N=1536 K=2048", 1536, 2048)):
    a = torch.randn(M, K, device="cuda").half()
    w = torch.randn(N, K, device="cuda") * 0.05
    hi = w.half(); lo = (w - hi.float()).half()
    out = torch.empty(M, N, dtype=torch.float32 if MODE in ("res", "f32") else torch.float16, device="cuda")
    if MODE == "res": out.zero_()
    f = (lambda: ops.gemm_nt(a, hi, out16=out, b_lo=lo, lo_scale=1.0)) if MODE == "split" else (lambda: ops.gemm_nt(a, hi, out16=out))
    if MODE == "res": f = lambda: ops.gemm_nt(a, hi, out32=out, res=out)        # the decoder's fc / conv2 form: f32 residual stream, in place
    if MODE == "f32": f = lambda: ops.gemm_nt(a, hi, out32=out)
    for _ in range(2): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for x, y in ev:
        x.record(); f(); y.record()
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in ev)[len(ev) // 2]
    print(f"{name}: {ms:.3f} ms  algorithmic {2.0 * M * N * K / ms / 1e9:.0f} TF/s  executed {EXE * M * N * K / ms / 1e9:.0f} TF/s   {MODE} lib={LIBSEL} m16={M32}")
    del a, w, hi, lo, out
