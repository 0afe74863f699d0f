#!/usr/bin/env python3
"""(Checker-side experiment.)  BASELINE configs[3]'s sampler at FULL length on a large-tile grid: 250 ancestral steps at
G = 72 (dithered weights live), tame family, un-clamped last x0 vs the CPU oracle - the 40-step version is a pytest
case (tests/test_gpu_engine.py::test_ddpm_large_grid_vs_oracle); this one takes 8.5 minutes of host time.
usage: python tests/tools/ddpm250_large_grid.py > profiles/<round>_ddpm250_g72.txt"""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import test_gpu_engine as T  # noqa: E402

print(T._ddpm_large_grid(250))
