#!/usr/bin/env python3
"""Time one-forward-at-a-time schedules: lanes x lane_graphs x batch.  usage: lane_sweep.py [f16|f32]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd"))
import numpy as np, torch
from whenet_hip import _lib, synth, weights as W

dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
blob = W.pack(W.synthetic(1234))
h = _lib.Handle(blob, device=0, dtype=_lib.F16 if dtype == "f16" else _lib.F32)
dev = torch.device("cuda:0")


def run(B, steps, **opts):
    for k, v in opts.items():
        h.set_option(k, v)
    crops = torch.from_numpy(synth.scene_crops(min(B, 64), seed=3)).to(dev)
    if B > 64:
        crops = crops.repeat((B + 63) // 64, 1, 1, 1)[:B].contiguous()
    y = torch.empty(B, 3, device=dev); a = torch.empty(B, 3, dtype=torch.int32, device=dev)
    l = torch.empty(B, 252, device=dev)
    def step():
        h.forward_device(crops.data_ptr(), B, y.data_ptr(), a.data_ptr(), l.data_ptr())
    for _ in range(5):
        step()
    h.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    h.sync(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # latency of one forward alone
    t1 = time.perf_counter()
    for _ in range(20):
        step(); h.sync()
    lat = (time.perf_counter() - t1) / 20
    return el / steps * 1e6, lat * 1e6, l.cpu().numpy().copy()

ref = {}
for B in (64, 16, 1, 512):
    for lanes, lg, mlc in ((1, 0, 16), (3, 0, 16), (2, 1, 8), (3, 1, 8), (4, 1, 8), (6, 1, 8), (8, 1, 8)):
        if lanes > 1 and B // lanes < 1:
            continue
        us, lat, lg_out = run(B, 100 if B <= 64 else 30, lanes=lanes, lane_graphs=lg, min_lane_crops=max(1, min(mlc, B // lanes)))
        same = "" if B not in ref else (" bits-equal" if np.array_equal(ref[B], lg_out) else " BITS DIFFER")
        ref.setdefault(B, lg_out)
        print(f"B={B:4d} lanes={lanes} lane_graphs={lg}: {us:8.1f} us/forward back-to-back ({B/us*1e6:9.0f} crops/s), alone {lat:8.1f} us{same}", flush=True)
