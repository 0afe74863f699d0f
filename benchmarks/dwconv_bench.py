import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; _lab.use_lab()   # the DVD_* switches exist in the lab build only (make -C dvd_amd/csrc lab)
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from dvd_amd import ops
n, side, c = 16, 144, 2048
x = torch.randn(n * side * side, c, device="cuda").half()
w = torch.randn(9, c, device="cuda"); b = torch.randn(c, device="cuda")
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
f = lambda: ops.dwconv3x3(x, w, b, n, side)
gb = 2.0 * n * side * side * c * 2 / 1e9          # f16 in + f16 out
for tx, ty in (("4", "1"), ("4", "2"), ("2", "2"), ("2", "4"), ("2", "8"), ("1", "4"), ("1", "8")):
    if True:
        os.environ["DVD_DWCONV_TY"], os.environ["DVD_DWCONV_TX"] = ty, tx
        ms = t(f)
        print(f"{tx} x {ty} tokens per thread: {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s")
del os.environ["DVD_DWCONV_TY"], os.environ["DVD_DWCONV_TX"]
for tx in ("2", "3", "4"):
    for sy in ("144", "72", "48", "36", "24", "16"):
        for nt in (False, True):
            os.environ["DVD_DWCONV_STRIP"], os.environ["DVD_DWCONV_SY"] = tx, sy
            os.environ["DVD_DWCONV_NT"] = "1" if nt else "0"
            ms = t(f)
            print(f"strip {tx} columns x {sy:>3} rows{' nt' if nt else '   '}: {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s")
for k in ("DVD_DWCONV_STRIP", "DVD_DWCONV_SY", "DVD_DWCONV_NT"): os.environ.pop(k, None)
ms = t(f)
print(f"product rule: {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s")
os.environ["DVD_DWCONV_V1"] = "1"
ms = t(f)
print(f"one token per thread (v1): {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s")
