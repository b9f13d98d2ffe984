#!/usr/bin/env python3
"""Phase timeline of the tail megakernel for one crop (wall clock, 10 ns ticks)."""
import os, sys
import numpy as np
import torch  # noqa
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd")); sys.path.insert(0, ROOT)
from whenet_hip import _lib, weights as W
dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
h = _lib.Handle(W.pack(W.synthetic(1234)), dtype=_lib.F16 if dt == "f16" else _lib.F32)
x = np.random.default_rng(0).normal(0, 1, (n, 14, 14, 80)).astype(np.float32)
r = h.op_tail(x)
t = r["timing"].astype(np.int64)
us = lambda a, b: (t[b] - t[a]) / 100.0
print(f"dtype {dt} n={n}: total {us(0, 91):.1f} us to decode start")
for bi in range(10):
    o = 1 + bi * 8
    prev = 0 if bi == 0 else 1 + (bi - 1) * 8 + 5
    print(f"b{7+bi:2d}: zeroE {us(prev, o):6.1f} | chunk0 expand {us(o, o+1):6.1f} dw {us(o+1, o+2):6.1f} | phase1 total {us(o, o+3):7.1f} | SE {us(o+3, o+4):6.1f} | project {us(o+4, o+5):6.1f}")
print(f"head conv+GAP {us(1 + 9*8 + 5, 90):.1f} us, dense {us(90, 91):.1f} us")
