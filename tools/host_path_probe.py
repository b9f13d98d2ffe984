#!/usr/bin/env python3
"""Host -> host throughput of the blocking entry points: get_angle(uint8[512]) through the drop-in class against the fan-out's staging
form / engine count / chunk size, and forward_u8 at 64 crops.  Run on the GPU box."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd"))
import torch  # noqa: F401  (pages the ROCm runtime in)
import whenet
from whenet_hip import _lib, weights as W

blob = W.pack(W.synthetic(1234))
rng = np.random.default_rng(0)
big = rng.integers(0, 256, (512, 224, 224, 3), dtype=np.uint8)


def rate(fn, n, secs=0.8, warm=4):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < secs:
        fn(); k += 1
    return k * n / (time.perf_counter() - t0)


for dd in ("f16", "f32s"):
    for inflight in (2, 3):
        with whenet.WHENet(snapshot=blob, dtype=dd, inflight=inflight) as m:
            for stage, chunk in ((0, 128), (1, 128), (2, 128), (2, 64), (2, 256)):
                m._handle.set_option("fanout_stage", stage)
                m._handle.set_option("fanout_chunk", chunk)
                r = rate(lambda: m.get_angle(big), 512)
                print(f"{dd} get_angle(uint8[512]) inflight={inflight} stage={stage} chunk={chunk}: {r / 1e3:7.1f} k crops/s", flush=True)
    with _lib.Handle(blob, device=0, dtype=_lib.F16 if dd == "f16" else _lib.F32S) as h:
        for n in (64, 128, 255):
            x = big[:n]
            r = rate(lambda: h.forward(x, want_logits=False), n, secs=0.5)
            print(f"{dd} forward_u8 blocking n={n}: {r / 1e3:7.1f} k crops/s", flush=True)
        fresh = rate(lambda: h.forward(big[:64].copy(), want_logits=False), 64, secs=0.5)
        print(f"{dd} forward_u8 blocking n=64, a fresh array per call: {fresh / 1e3:7.1f} k crops/s", flush=True)
with whenet.WHENet(snapshot=blob, dtype="f16") as m:            # the class as the bench's dropin leg uses it (lazy fan-out)
    r = rate(lambda: m.get_angle(big), 512)
    r2 = rate(lambda: m.get_angle(big.copy()), 512)
    print(f"f16 default class get_angle(uint8[512]): {r / 1e3:7.1f} k crops/s; a fresh array per call {r2 / 1e3:7.1f} k (incl. the 77 MB numpy copy)")
