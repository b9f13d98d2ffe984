#!/usr/bin/env python3
"""B = 1 / 2 / 4 latency (device-resident and host call) against option se_fuse_tiny."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd"))
import torch
from whenet_hip import _lib, weights as W
blob = W.pack(W.synthetic(1234))
crops = torch.randint(0, 256, (8, 224, 224, 3), dtype=torch.uint8, device="cuda")
host = crops.cpu().numpy()
y = torch.empty((8, 3), device="cuda"); a = torch.empty((8, 3), dtype=torch.int32, device="cuda"); l = torch.empty((8, 252), device="cuda")
for name, dt in (("f32", _lib.F32), ("f32s", _lib.F32S), ("f16", _lib.F16)):
    h = _lib.Handle(blob, device=0, dtype=dt)
    ref = None
    for tiny in (0, 4, 0, 4):
        h.set_option("se_fuse_tiny", tiny)
        out = []
        for n in (1, 2, 4):
            lat = []
            for i in range(500):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                h.forward_device(crops.data_ptr(), n, y.data_ptr(), a.data_ptr(), l.data_ptr()); h.sync()
                lat.append(time.perf_counter() - t0)
            dev = np.median(lat[100:]) * 1e6
            lat = []
            for i in range(400):
                t0 = time.perf_counter(); r = h.forward(host[:n], want_logits=True); lat.append(time.perf_counter() - t0)
            out.append((n, dev, np.median(lat[100:]) * 1e6))
            if n == 4:
                if ref is None: ref = r
                else: assert all(np.array_equal(u, v) for u, v in zip(ref, r)), "se_fuse_tiny changes bits"
        print(f"{name} se_fuse_tiny={tiny} launches {h.info().n_kernels_per_forward}: " + "  ".join(f"B={n}: device {d:.1f} us, host call {w:.1f} us" for n, d, w in out), flush=True)
    h.close()
