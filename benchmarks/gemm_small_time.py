#!/usr/bin/env python3
"""The sampler's per-step GEMMs at the reference's native point (G = 64: 1024 tokens x 2 hypotheses = 2048 rows per document,
(hi, lo) weight pairs, 128 x 128 tiles): what the library dispatches (product) beside the register-staged 4-wave kernel (w4), the LDS-DMA ring kernel where eligible
(ring: 128 x 128 tiles; r256: 128 x 256 tiles, eight waves) - `--lab` switches DVD_GEMM_RING128 / _RING256 (also _W8 / _PD4), interleaved in one process.  usage: python benchmarks/gemm_small_time.py [docs=1] [reps=20] --lab"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; LIBSEL = _lab.which()
import torch
from dvd_amd import ops
args = [a for a in sys.argv[1:] if not a.startswith("--")]
docs = int(args[0]) if len(args) > 0 else 1
reps = int(args[1]) if len(args) > 1 else 20
M = 2048 * docs
small = 2 if M >= 16384 else 1


def timed(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for x, y in ev:
        x.record(); f(); y.record()
    torch.cuda.synchronize()
    return sorted(x.elapsed_time(y) for x, y in ev)[len(ev) // 2] * 1e3


tot = {}
VARIANTS = ("product", "w4", "ring", "r256")     # product = what the library dispatches; pd4 = lab switch.  (rd0/1/2 in profiles/r5_gemm_small_variants.txt
# were fragment-read placements of a lab build that was not kept: all 16 reads of a K-tile first = +8 %, slower.)
for name, N, K, f32out in (("qk  N=3072 K=1536", 3072, 1536, False), ("c1  N=2048 K=1536", 2048, 1536, False),
                           ("fc  N=1536 K=1536", 1536, 1536, True), ("c2  N=1536 K=2048", 1536, 2048, True),
                           ("dit N=384  K=384 ", 384, 384, True), ("dit N=1536 K=384 ", 1536, 384, False),
                           ("f32 N=256 K=2304 ", 256, 2304, None)):
    if f32out is None:                      # the conv pyramid's exact-f32 shape family (M = 4096 pixels per document)
        a = torch.randn(4096 * docs, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05
        out = torch.empty(a.shape[0], N, device="cuda")
        f = lambda: ops.gemm_nt(a, w, out32=out)
        flops = 2.0 * a.shape[0] * N * K
    else:
        a = torch.randn(M, K, device="cuda").half()
        w = torch.randn(N, K, device="cuda") * 0.05
        hi = w.half(); lo = ((w - hi.float()) * 2048.0).half()
        out = torch.empty(M, N, dtype=torch.float32 if f32out else torch.float16, device="cuda")
        kw = dict(out32=out) if f32out else dict(out16=out)
        f = lambda: ops.gemm_nt(a, hi, b_lo=lo, small_tiles=small, **kw)
        flops = 4.0 * M * N * K
    res = {}
    for tag in VARIANTS * 2:
        for k in ("DVD_GEMM_PD4", "DVD_GEMM_W8", "DVD_GEMM_RING128", "DVD_GEMM_RING256"): os.environ.pop(k, None)
        if tag != "product": os.environ["DVD_GEMM_RING256"] = "2" if tag == "r256" else "0"
        if tag == "pd4": os.environ["DVD_GEMM_PD4"], os.environ["DVD_GEMM_RING128"] = "1", "0"
        if tag in ("w4", "w8"): os.environ["DVD_GEMM_W8"], os.environ["DVD_GEMM_RING128"] = ("1" if tag == "w8" else "0"), "0"
        if tag == "ring": os.environ["DVD_GEMM_RING128"] = "1"
        res.setdefault(tag, []).append(timed(f))
    for k in ("DVD_GEMM_PD4", "DVD_GEMM_W8", "DVD_GEMM_RING128", "DVD_GEMM_RING256"): os.environ.pop(k, None)
    best = {t: min(v) for t, v in res.items()}
    for t, v in best.items(): tot[t] = tot.get(t, 0.0) + v
    print(f"{name} rows {a.shape[0]:6d}: " + "  ".join(f"{t} {v:7.1f} us" for t, v in best.items()) +
          f"   ({flops / best['product'] / 1e6:6.0f} TF/s executed, product)   lib={LIBSEL}")
print("sum: " + "  ".join(f"{t} {v:.1f} us" for t, v in tot.items()))
