#!/usr/bin/env python3
"""Several device-resident forwards in flight (option inflight=3), every round compared bit for bit with a serial
reference, batch 64 and 512, with and without option fold12."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from whenet_hip import _lib, synth, weights as W
blob = W.pack(W.synthetic(1234))
dev = torch.device("cuda:0")
for B in (64, 512):
    crops = torch.from_numpy(synth.noise_crops(B, seed=0)).to(dev)
    for fold in (1, 0):
        h = _lib.Handle(blob, device=0, dtype=_lib.F16)
        h.set_option("fold12", fold)
        M = 3
        outs = [(torch.zeros((B, 3), dtype=torch.float32, device=dev), torch.zeros((B, 3), dtype=torch.int32, device=dev),
                 torch.zeros((B, 252), dtype=torch.float32, device=dev)) for _ in range(M)]
        h.forward_device(crops.data_ptr(), B, *[t.data_ptr() for t in outs[0]]); h.sync(); torch.cuda.synchronize()
        ref = outs[0][2].clone()
        h.set_option("inflight", M)
        bad = 0; rounds = 400 if B == 64 else 80
        for r in range(rounds):
            for i in range(5):
                y, a, l = outs[i % M]
                h.forward_device(crops.data_ptr(), B, y.data_ptr(), a.data_ptr(), l.data_ptr())
            h.sync(); torch.cuda.synchronize()
            for s in range(M):
                if not torch.equal(outs[s][2], ref):
                    bad += 1
                    d = torch.nonzero((outs[s][2] != ref).any(dim=1)).flatten().tolist()
                    if bad <= 5:
                        print(f"  round {r} slot {s}: {len(d)} crops differ {d[:12]} max {float((outs[s][2]-ref).abs().max()):.4g}")
        print(f"B={B} fold12={fold}: {bad} bad slot-results in {rounds} rounds")
        h.close()
