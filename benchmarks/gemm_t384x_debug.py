#!/usr/bin/env python3
"""Round 6 debug aid: where do the outputs of the 16x16x32 t384 loop land?  A = rows of small integers, B = identity-like, exact
comparison with the integer product; prints the first mismatching rows / columns by 16-blocks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; _lab.use_lab()
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from dvd_amd import ops
M, N, K = 384, 256, 256
rng = np.random.RandomState(0)
a = torch.from_numpy(rng.randint(-4, 5, (M, K)).astype(np.float32)).half().cuda()
b = torch.from_numpy(rng.randint(-4, 5, (N, K)).astype(np.float32)).half().cuda()
ref = (a.float() @ b.float().t()).cpu()
for mode in ("f32", "f16"):
    if mode == "f32":
        out = torch.full((M, N), 7.0, device="cuda"); ops.gemm_nt(a, b, out32=out)
    else:
        out = torch.full((M, N), 7.0, dtype=torch.float16, device="cuda"); ops.gemm_nt(a, b, out16=out)
    bad = (out.float().cpu() != ref)
    print(mode, "mismatches", int(bad.sum()), "of", bad.numel())
    if bad.any():
        blk = bad.reshape(M // 16, 16, N // 16, 16).any(3).any(1)
        print("bad 16x16 blocks (rows x cols):"); print(blk.int().numpy()[:12, :16])
        # is the data a permutation?  find for out[r0, c0] the matching ref position in the same 96 x 128 wave tile
        o = out.float().cpu()
        for (r0, c0) in ((0, 0), (1, 0), (0, 1), (4, 0), (0, 16), (16, 0), (8, 3)):
            hits = (ref[:96, :128] == o[r0, c0]).nonzero()[:4].tolist()
            print(f"out[{r0},{c0}] = {o[r0, c0].item()}  ref[{r0},{c0}] = {ref[r0, c0].item()}  found at {hits}")
