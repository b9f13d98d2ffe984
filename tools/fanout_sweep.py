#!/usr/bin/env python3
"""get_angle-shaped host->host throughput of ONE large blocking call (whenet_forward_u8, pageable numpy input) against the
fan-out knobs of capi.cpp (inflight engines x chunk x staging mode x depth), plus the drop-in's B=1 wall latency.
Run on the GPU box:  python tools/fanout_sweep.py [f16|f32]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd"))
import torch  # noqa: F401,E402
from whenet_hip import _lib, synth, weights as W  # noqa: E402


def rate(fn, n, secs=0.6):
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < secs:
        fn()
        k += 1
    return k * n / (time.perf_counter() - t0)


def main():
    dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
    dtype = _lib.F16 if dt == "f16" else _lib.F32
    blob = W.pack(W.synthetic(1234))
    rng = np.random.default_rng(0)
    big = rng.integers(0, 256, (1024, 224, 224, 3), dtype=np.uint8)
    out = {"dtype": dt, "rows": []}
    h = _lib.Handle(blob, device=0, dtype=dtype)
    for N in (128, 256, 512, 1024):
        crops = big[:N]
        h.set_option("inflight", 1)
        h.set_option("fanout_min", 0)
        base = rate(lambda: h.forward(crops, want_logits=False), N)
        out["rows"].append({"N": N, "mode": "one forward (no fan-out)", "crops_s": base})
        print(f"N={N:5d} plain blocking forward: {base:9.0f} crops/s", flush=True)
        for inflight, chunk, stage, depth in ((1, 128, 0, 2), (1, 128, 1, 2), (1, 256, 1, 2), (2, 64, 0, 2), (2, 64, 1, 2), (3, 64, 0, 2), (3, 64, 1, 2),
                                              (3, 64, 1, 1), (3, 64, 1, 3), (4, 64, 1, 2), (3, 96, 1, 2), (3, 128, 1, 2), (2, 128, 1, 2), (3, 32, 1, 2)):
            if chunk >= N:
                continue
            h.set_option("inflight", inflight)
            h.set_option("fanout_min", 128)
            h.set_option("fanout_chunk", chunk)
            h.set_option("fanout_stage", stage)
            h.set_option("fanout_depth", depth)
            r = rate(lambda: h.forward(crops, want_logits=False), N)
            out["rows"].append({"N": N, "inflight": inflight, "chunk": chunk, "stage": stage, "depth": depth, "crops_s": r})
            print(f"N={N:5d} inflight={inflight} chunk={chunk:3d} stage={stage} depth={depth}: {r:9.0f} crops/s", flush=True)
    h.close()
    # the drop-in class itself (python overhead included)
    import whenet
    for dd in ("f16", "f32"):
        m = whenet.WHENet(dtype=dd)
        c512, c1 = big[:512], big[:1]
        r = rate(lambda: m.get_angle(c512), 512, 1.0)
        lat = []
        for i in range(600):
            a = time.perf_counter()
            m.get_angle(c1)
            lat.append(time.perf_counter() - a)
        lat = np.array(lat[100:]) * 1e6
        m._handle.set_option("se_fuse", 0)
        lat0 = []
        for i in range(400):
            a = time.perf_counter()
            m.get_angle(c1)
            lat0.append(time.perf_counter() - a)
        lat0 = np.array(lat0[100:]) * 1e6
        row = {"dropin": dd, "get_angle_b512_crops_s": r, "get_angle_b1_median_us": float(np.median(lat)), "get_angle_b1_p99_us": float(np.percentile(lat, 99)),
               "get_angle_b1_median_us_se_fuse0": float(np.median(lat0))}
        out["rows"].append(row)
        print(json.dumps(row), flush=True)
        m.close()
    with open(os.path.join(ROOT, "gpurun_out", f"fanout_sweep_{dt}.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
