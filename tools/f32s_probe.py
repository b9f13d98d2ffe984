#!/usr/bin/env python3
"""f32 vs f32s: device-resident throughput at 64 crops (1 and 3 forwards in flight) and per-layer times."""
import os, sys, time, json
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd"))
import torch
from whenet_hip import _lib, weights as W

blob = W.pack(W.synthetic(1234))
B = 64
crops = torch.randint(0, 256, (B, 224, 224, 3), dtype=torch.uint8, device="cuda")
outs = [(torch.empty((B, 3), device="cuda"), torch.empty((B, 3), dtype=torch.int32, device="cuda"), torch.empty((B, 252), device="cuda")) for _ in range(4)]
for name, dt in (("f32", _lib.F32), ("f32s", _lib.F32S), ("f16", _lib.F16)):
    h = _lib.Handle(blob, device=0, dtype=dt)
    for inflight in (1, 3):
        h.set_option("inflight", inflight)
        def step(i):
            o = outs[i % inflight]
            h.forward_device(crops.data_ptr(), B, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
        for i in range(12): step(i)
        h.sync()
        t0 = time.perf_counter(); K = 60
        for i in range(K): step(i)
        h.sync()
        dt_s = time.perf_counter() - t0
        print(f"{name} B={B} inflight={inflight}: {K * B / dt_s:9.0f} crops/s", flush=True)
    h.set_option("inflight", 1)
    if name != "f16":
        prof = h.profile(crops.data_ptr(), B, 5)
        agg = {}
        for p in prof:
            agg.setdefault(p["kind"], 0.0); agg[p["kind"]] += p["avg_us"]
        print(name, "per-kind us (one chain per lane):", {k: round(v, 1) for k, v in agg.items()}, flush=True)
        for p in prof:
            if p["kind"] in ("pw", "front"): print(f"   {p['layer']:14s} {p['avg_us']:7.1f} us  {p['kernel']}", flush=True)
    h.close()
