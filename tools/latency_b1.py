#!/usr/bin/env python3
"""Batch-1 (and batch-8) forward latency, f32 and f16, under engine options given as KEY=VALUE groups on the
command line (one measurement per argument; options inside a group separated by commas).
usage: python tools/latency_b1.py "se_fuse=0" "se_fuse=1" "se_fuse=2,front_impl=0" """
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from whenet_hip import _lib, synth, weights as W
blob = W.pack(W.synthetic(1234))
dev = torch.device("cuda", 0)
groups = sys.argv[1:] or [""]
for nb in (1, 8):
    crops = torch.from_numpy(synth.noise_crops(nb, seed=3)).to(dev)
    y = torch.zeros((nb, 3), device=dev); a = torch.zeros((nb, 3), dtype=torch.int32, device=dev); l = torch.zeros((nb, 252), device=dev)
    for name, dt in (("f32", _lib.F32), ("f16", _lib.F16)):
        for grp in groups:
            h = _lib.Handle(blob, device=0, dtype=dt)
            for kv in filter(None, grp.split(",")):
                k, v = kv.split("=")
                h.set_option(k, int(v))
            lat = []
            for i in range(700):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                h.forward_device(crops.data_ptr(), nb, y.data_ptr(), a.data_ptr(), l.data_ptr())
                h.sync()
                lat.append(time.perf_counter() - t0)
            lat = np.array(lat[100:]) * 1e6
            print(f"batch {nb} {name} [{grp or 'defaults':24s}]: median {np.median(lat):7.1f} us  p99 {np.percentile(lat, 99):7.1f} us  "
                  f"({h.info().n_kernels_per_forward} launches)")
            h.close()
