#!/usr/bin/env python3
"""A/B of option pw_staged (split-K GEMMs: activation rows fetched coalesced and staged through LDS) on the same box:
throughput at B=64 (1 and 3 forwards in flight) and B=512, per-layer times of the project convs, bits against each other."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd"))
import torch
from whenet_hip import _lib, synth, weights as W

blob = W.pack(W.synthetic(1234))
which = sys.argv[1:] or ["f16", "f32", "f32s"]
DT = {"f16": _lib.F16, "f32": _lib.F32, "f32s": _lib.F32S}
host = synth.scene_crops(64, seed=3)
for name in which:
    res = {}
    for staged in (0, 1, 0, 1):
        h = _lib.Handle(blob, device=0, dtype=DT[name])
        h.set_option("pw_staged", staged)
        y, a, l = h.forward(host)
        res.setdefault(staged, (y, l))
        for B in (1, 8, 64, 512):
            crops = torch.randint(0, 256, (B, 224, 224, 3), dtype=torch.uint8, device="cuda")
            outs = [(torch.empty((B, 3), device="cuda"), torch.empty((B, 3), dtype=torch.int32, device="cuda"), torch.empty((B, 252), device="cuda")) for _ in range(3)]
            for inflight in ((1, 3) if B <= 64 else (1,)):
                h.set_option("inflight", inflight)
                def step(i):
                    o = outs[i % inflight]
                    h.forward_device(crops.data_ptr(), B, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
                for i in range(9): step(i)
                h.sync()
                K = {1: 300, 8: 200, 64: 90, 512: 15}[B]
                t0 = time.perf_counter()
                for i in range(K): step(i)
                h.sync()
                print(f"{name} pw_staged={staged} B={B} inflight={inflight}: {K * B / (time.perf_counter() - t0):9.0f} crops/s", flush=True)
            h.set_option("inflight", 1)
        if staged in (0, 1) and len(res) <= 2:
            crops = torch.randint(0, 256, (64, 224, 224, 3), dtype=torch.uint8, device="cuda")
            h.set_option("lanes", 1)
            prof = h.profile(crops.data_ptr(), 64, 8)
            print("   ", name, f"pw_staged={staged}", "split-K layers, one chain of 64 crops (us):",
                  " ".join(f"{p['layer'].split('/')[0]}:{p['avg_us']:.1f}" for p in prof if "splitk" in p["kernel"]),
                  "| chain", round(sum(p["avg_us"] for p in prof), 1), flush=True)
        h.close()
    d = np.abs(res[0][0] - res[1][0]).max()
    print(f"{name}: max |angle(staged) - angle(direct)| = {d:.3e} deg, logits {np.abs(res[0][1] - res[1][1]).max():.3e}", flush=True)
