#!/usr/bin/env python3
"""Repeat-run stress: many forwards of ragged batches (all options that involve inter-workgroup hand-offs) must
reproduce the first result bit for bit.  usage: stress_repeat.py [iters]"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd"))
import numpy as np, torch
from whenet_hip import _lib, synth, weights as W

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
blob = W.pack(W.synthetic(1234))
dev = torch.device("cuda:0")
bad = 0
for dtype in (_lib.F16, _lib.F32):
    h = _lib.Handle(blob, device=0, dtype=dtype)
    h.set_option("inflight", 3)
    for B in (1, 7, 64, 150):
        crops = torch.from_numpy(synth.scene_crops(min(B, 40), seed=B)).to(dev)
        if B > 40:
            crops = crops.repeat((B + 39) // 40, 1, 1, 1)[:B].contiguous()
        outs = [(torch.zeros(B, 3, device=dev), torch.zeros(B, 3, dtype=torch.int32, device=dev), torch.zeros(B, 252, device=dev))
                for _ in range(3)]
        ref = None
        for i in range(iters):
            y, a, l = outs[i % 3]
            h.forward_device(crops.data_ptr(), B, y.data_ptr(), a.data_ptr(), l.data_ptr())
            if i % 3 == 2 or i == iters - 1:
                h.sync()
                for (yy, aa, ll) in outs:
                    if ref is None:
                        ref = (yy.clone(), aa.clone(), ll.clone())
                    elif ll.abs().sum() > 0 and not (torch.equal(ll, ref[2]) and torch.equal(yy, ref[0]) and torch.equal(aa, ref[1])):
                        bad += 1
        print(f"dtype {dtype} B={B}: {iters} forwards, mismatches so far {bad}", flush=True)
    h.close()
print("STRESS", "OK" if bad == 0 else f"FAILED ({bad})")
sys.exit(1 if bad else 0)
