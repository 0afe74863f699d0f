#!/usr/bin/env python3
"""Lab check of the one-tensor mode of the 3-stage 256x256 GEMM (DVD_GEMM_RING=1, lab build) against float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; _lab.use_lab()
import torch
from dvd_amd import ops
os.environ["DVD_GEMM_RING"] = "1"
for M, N, K in ((1024, 512, 256), (700, 256, 1536), (4096, 1536, 2048)):
    a = torch.randn(M, K, device="cuda").half(); b = (torch.randn(N, K, device="cuda") * 0.05).half()
    out = torch.empty(M, N, device="cuda")
    ops.gemm_nt(a, b, out32=out)
    ref = a.double() @ b.double().t()
    print(M, N, K, "max err", float((out.double() - ref).abs().max()), "ref max", float(ref.abs().max()))
