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
print("row4 ms", t(f))
os.environ["DVD_DWCONV_V1"] = "1"
print("v1   ms", t(f))
