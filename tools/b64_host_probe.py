#!/usr/bin/env python3
"""The drop-in's mid-size blocking calls (N = 16..192): one forward against the chunked fan-out with small chunks."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd"))
import torch
from whenet_hip import _lib, weights as W

def rate(fn, n, secs=0.35):
    for _ in range(3): fn()
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < secs:
        fn(); k += 1
    return k * n / (time.perf_counter() - t0)

blob = W.pack(W.synthetic(1234))
big = np.random.default_rng(0).integers(0, 256, (192, 224, 224, 3), dtype=np.uint8)
for name, dt in (("f16", _lib.F16), ("f32s", _lib.F32S)):
    h = _lib.Handle(blob, device=0, dtype=dt)
    for N in (16, 32, 64, 96, 128, 192):
        crops = big[:N]
        h.set_option("inflight", 1); h.set_option("fanout_min", 0)
        row = [f"plain {rate(lambda: h.forward(crops, want_logits=True), N) / 1e3:6.1f}k"]
        for infl in (2, 3, 4):
            h.set_option("inflight", infl)
            for stage in (0, 1):
                h.set_option("fanout_stage", stage)
                chunk = max(8, (N + infl - 1) // infl)
                h.set_option("fanout_min", N); h.set_option("fanout_chunk", chunk)
                row.append(f"{infl}x{chunk}s{stage} {rate(lambda: h.forward(crops, want_logits=True), N) / 1e3:6.1f}k")
        print(f"{name} N={N:3d}: " + "  ".join(row), flush=True)
    h.close()
