#!/usr/bin/env python3
"""3x3 convolutions of the once-per-document stage as the library runs them (dvd_conv3x3_nhwc: implicit GEMMs, exact-f32 MFMA),
at the pyramid's and the nets' shapes: microseconds and TFLOP/s against the 157 TF/s f32 matrix peak.
usage: python benchmarks/conv_time.py [reps=10] [--lab]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _lab; LIBSEL = _lab.which()
import torch
from dvd_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for name, cin, cout, h, w in (("pyramid 64->64 @512", 64, 64, 512, 512), ("pyramid 64->128 @256", 64, 128, 256, 256),
                              ("pyramid 128->128 @256", 128, 128, 256, 256), ("pyramid 256->256 @128", 256, 256, 128, 128),
                              ("net 64->64 @288 x8", 64, 64, 288 * 8, 288), ("net 64->64 @288", 64, 64, 288, 288),
                              ("net 16->16 @288 x8", 16, 16, 288 * 8, 288), ("net 32->16 @288 x8", 32, 16, 288 * 8, 288),
                              ("net 16->16 @288", 16, 16, 288, 288), ("net 64->16 @144", 64, 16, 144, 144),
                              ("net 16->16 @36", 16, 16, 36, 36), ("net 32->16 @18", 32, 16, 18, 18), ("net 128->64 @36", 128, 64, 36, 36)):
    x = torch.randn(h * w, cin, device="cuda")
    wp = torch.randn(cout, 9 * cin, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda")
    f = lambda: ops.conv3x3_relu_nhwc_implicit(x, wp, b, cin, cout, h, w)
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    us = sorted(s.elapsed_time(e) for s, e in ev)[len(ev) // 2] * 1e3
    fl = 2.0 * h * w * cout * 9 * cin
    print(f"{name:24s} rows {h * w:8d}  K {9 * cin:5d}  N {cout:4d}: {us:9.1f} us  {fl / us / 1e6:7.1f} TF/s   lib={LIBSEL}")
