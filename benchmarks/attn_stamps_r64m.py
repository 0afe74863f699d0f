#!/usr/bin/env python3
"""Where a workgroup of flash_attn_r64x_kernel (or, with `r64m`, of its 32x32x16 sibling) spends its cycles: s_memtime at kernel entry, loop entry, loop exit and
kernel exit (four stamps per wave, all OUTSIDE the tile loop - the loop runs unperturbed), and the dispatch picture: how
many workgroups each CU ran and the gaps between one workgroup's exit and the next one's entry on the same CU.
usage: [B=16] [ablation] [r64m]   (lab build; default: the production kernel r64x)"""
import os, sys
if len(sys.argv) > 3 and sys.argv[3] == "r64m":     # the 32x32x16 sibling (same stamps); ablation numbers are then r64m's
    os.environ["DVD_ATTN_R64M"] = "1"
    if sys.argv[2] != "0":
        os.environ["DVD_ATTN_R64M_ABL"] = sys.argv[2]
elif len(sys.argv) > 2 and sys.argv[2] != "0":
    os.environ["DVD_ATTN_R64X_ABL"] = sys.argv[2]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; _lab.use_lab()
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import ctypes as C
import torch
from dvd_amd import lib, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
hd, T = 256, 20736
Cc = 6 * hd
qk = torch.randn(B, T, 2 * Cc, device="cuda").half(); vt = torch.randn(B, Cc, T, device="cuda").half()
out = torch.empty(B, T, Cc, dtype=torch.float16, device="cuda")
nwg = (T // 256) * 6 * B
st = torch.zeros(nwg * 4 * 6, dtype=torch.int64, device="cuda")
for _ in range(2):
    ops.flash_attn(qk[:, :, :Cc], qk[:, :, Cc:], vt, out, 6, hd, 1.0 / (hd ** 0.5))
lib.call("dvd_attn_debug_stamps", C.c_void_p(st.data_ptr()))
ops.flash_attn(qk[:, :, :Cc], qk[:, :, Cc:], vt, out, 6, hd, 1.0 / (hd ** 0.5))
torch.cuda.synchronize()
lib.call("dvd_attn_debug_stamps", C.c_void_p(0))
s = st.view(nwg, 4, 6).cpu()
d = s[:, :, :3].double()
nt = T // 32
print(f"per wave: prologue {d[..., 0].mean():9.0f} cycles   loop {d[..., 1].mean():10.0f} = {d[..., 1].mean() / nt:7.1f} per tile "
      f"(min wave {d[..., 1].min() / nt:.1f}, max {d[..., 1].max() / nt:.1f}; MFMA minimum 2048)   epilogue {d[..., 2].mean():8.0f}")
tot = d.sum(-1).mean()
print(f"workgroup = {tot:.0f} cycles: prologue {100 * d[..., 0].mean() / tot:.2f} %, loop {100 * d[..., 1].mean() / tot:.2f} %, epilogue {100 * d[..., 2].mean() / tot:.2f} %")
t0, t1 = s[:, :, 3].min(1).values, s[:, :, 4].max(1).values      # workgroup = first wave in .. last wave out
hw = s[:, 0, 5]
# gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID in bits 35:32 here.  s_memtime counters are per XCD: only
# stamps of the SAME CU are compared.
cu = ((hw >> 32) & 0xf) * 65536 + (hw & 0xff00)
ids, cnt = torch.unique(cu, return_counts=True)
print(f"{len(ids)} CUs seen; workgroups per CU: " + str(dict(zip(*[x.tolist() for x in torch.unique(cnt, return_counts=True)]))))
gaps, spans, inside = [], [], []
for i in ids.tolist():
    m = cu == i
    a, b = t0[m], t1[m]
    o = torch.argsort(a)
    a, b = a[o], b[o]
    gaps.append((a[1:] - b[:-1]).double())
    spans.append(float(b[-1] - a[0]))
    inside.append(float((b - a).sum()))
g = torch.cat(gaps)
print(f"gap between a workgroup's last exit and the next workgroup's first entry on the same CU: mean {g.mean():.0f}, median {g.median():.0f}, "
      f"p95 {g.quantile(0.95):.0f}, max {g.max():.0f} cycles")
sp = torch.tensor(spans)
print(f"per-CU span first entry .. last exit: mean {sp.mean():.0f}, min {sp.min():.0f}, max {sp.max():.0f} cycles; inside workgroups "
      f"{100 * sum(inside) / sp.sum().item():.2f} % of it")
