#!/usr/bin/env python3
"""Phase timeline of the trunk kernel (trunk.hip) for the crop of cluster 0, member 0 (wall clock,
10 ns ticks).  usage: trunk_timeline.py [f16|f32] [n] [key=value engine options ...]"""
import os, sys
import numpy as np
import torch  # noqa
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd")); sys.path.insert(0, ROOT)
from whenet_hip import _lib, weights as W
dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
h = _lib.Handle(W.pack(W.synthetic(1234)), dtype=_lib.F16 if dt == "f16" else _lib.F32)
tb = 3
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    h.set_option(k, int(v))
    if k == "trunk_timing_block":
        tb = int(v)
x = np.random.default_rng(0).normal(0, 1, (n, 14, 14, 80)).astype(np.float32)
r = h.op_trunk(x)
t = r["timing"].astype(np.int64)
us = lambda a, b: (t[b] - t[a]) / 100.0
print(f"dtype {dt} n={n} {' '.join(sys.argv[3:])}: total {us(0, 91):.1f} us; shader clock {(t[93] - t[92]) / max(us(0, 91), 1e-9):.0f} MHz (clock64 ticks / wall us)")
names = ("wait3+gather", "expand0", "dw0", "rest1", "se1+arrive+prefetch+wait1", "r+gate+stage", "project+arrive2+wait2", "reduce+arrive3")
tot = {k: 0.0 for k in names}
for bi in range(10):
    o = 1 + bi * 8
    prev = 0 if bi == 0 else 1 + (bi - 1) * 8 + 7
    vals = [us(prev, o)] + [us(o + i, o + i + 1) for i in range(7)]
    for k, v in zip(names, vals):
        tot[k] += v
    print(f"b{7+bi:2d}: " + " | ".join(f"{v:5.1f}" for v in vals) + f" | block {us(prev, o + 7):6.1f}")
print("cols: " + " | ".join(names))
print("sum: " + " | ".join(f"{v:6.1f}" for v in tot.values()))
print(f"head: wait+gather {us(1 + 9*8 + 7, 88):.1f} conv+GAP {us(88, 89):.1f} dense+sync {us(89, 90):.1f} decode {us(90, 91):.1f}")
d = t[128:]
dn = ["start", "bias+zeroE+taps0", "wait3", "gather", "expand0", "it0", "it1", "it2", "it3", "it4", "it5", "it6", "it7",
      "colsum+w1 issue", "se1", "arrive1", "issue Wp/D/w2c", "wait1", "r", "gate", "stage D,W*g", "project", "arrive2", "wait2",
      "reduce", "arrive3"]
prev = d[0]
out = []
for i in range(1, 26):
    if d[i] == 0:
        continue
    out.append(f"{dn[i]} {(d[i] - prev) / 100.0:.2f}")
    prev = d[i]
print(f"detail b{7+tb}: " + " | ".join(out) + f" | total {(prev - d[0]) / 100.0:.1f}")
