#!/usr/bin/env python3
"""Structured inputs that separate the parts of a head_dim-256 attention kernel variant (lab build):
V = 1 (row sums vs PV), K = 0 (V^T image and the O^T map), one-hot keys (key order), full random vs float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; _lab.use_lab()
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from dvd_amd import ops
for e in sys.argv[1:]:
    k, _, v = e.partition(":"); os.environ[k] = v or "1"
torch.manual_seed(0)
B, H, hd, T = 1, 1, 256, 256
C = H * hd
def run(q, k, v):
    out = torch.empty(B, T, C, dtype=torch.float16, device="cuda")
    ops.flash_attn(q.half().contiguous(), k.half().contiguous(), v.transpose(1, 2).half().contiguous(), out, H, hd, 1.0 / 16)
    torch.cuda.synchronize()
    return out.float()
def ref(q, k, v):
    s = (q.double() @ k.double().transpose(1, 2)) / 16
    return (torch.softmax(s, -1) @ v.double()).float()
dev = "cuda"
q = torch.randn(B, T, C, device=dev); k = torch.randn(B, T, C, device=dev); v = torch.randn(B, T, C, device=dev)
def report(name, o, r):
    d = (o - r).abs()
    print(f"{name:40s} max err {d.max().item():.3e}  mean {d.mean().item():.3e}  (ref max {r.abs().max().item():.3f})  nan {torch.isnan(o).sum().item()}")
    return d
d = report("V = 1", run(q, k, torch.ones_like(v)), torch.ones(B, T, C, device=dev))
d = report("K = 0 (uniform P)", run(q, torch.zeros_like(k), v), ref(q, torch.zeros_like(k), v))
if d.max() > 1e-2:
    dd = d[0]
    print("   worst rows", torch.topk(dd.max(1).values, 8).indices.tolist(), " worst dims", torch.topk(dd.max(0).values, 8).indices.tolist())
# V[key] = key index in dim 0..255 identical -> out[q] = sum_k P[q,k] * k: with a sharp softmax this reads off WHICH key each P pairs with
vk = torch.arange(T, device=dev, dtype=torch.float32).view(1, T, 1).expand(B, T, C).contiguous() / 16
d = report("random Q K, V[key] = key / 16", run(q, k, vk), ref(q, k, vk))
d = report("random", run(q, k, v), ref(q, k, v))
if d.max() > 1e-2:
    dd = d[0]
    print("   error by (row % 64): ", [round(x, 3) for x in dd.view(T // 64, 64, C).amax((0, 2)).tolist()][:64])
    print("   error by (dim % 16): ", [round(x, 3) for x in dd.view(T, C // 16, 16).amax((0, 1)).tolist()])
