#!/usr/bin/env python3
"""B=1 device-resident latency (median / p99 over 600 calls) per dtype and option set.  usage: latency_ab.py "k=v k=v" "k=v" ..."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd")); sys.path.insert(0, ROOT)
from whenet_hip import _lib, synth, weights as W
blob = W.pack(W.synthetic(1234))
dev = torch.device("cuda", 0)
crops = torch.from_numpy(synth.noise_crops(1, seed=3)).to(dev)
y = torch.zeros((1, 3), dtype=torch.float32, device=dev); a = torch.zeros((1, 3), dtype=torch.int32, device=dev); l = torch.zeros((1, 252), dtype=torch.float32, device=dev)
sets = sys.argv[1:] or ["-"]
for name, dt in (("f16", _lib.F16), ("f32s", _lib.F32S), ("f32", _lib.F32)):
    for st in sets:
        h = _lib.Handle(blob, device=0, dtype=dt)
        if st != "-":
            for kv in st.split():
                k, v = kv.split("="); h.set_option(k, int(v))
        lat = []
        for i in range(700):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h.forward_device(crops.data_ptr(), 1, y.data_ptr(), a.data_ptr(), l.data_ptr()); h.sync()
            lat.append(time.perf_counter() - t0)
        lat = np.array(lat[100:]) * 1e6
        print(f"B=1 {name} [{st}]: median {np.median(lat):.1f} us  p99 {np.percentile(lat, 99):.1f} us  launches {h.info().n_kernels_per_forward}", flush=True)
        h.close()
