#!/usr/bin/env python3
"""Round 5: (1) repeated trials of the fan-out candidates (the first sweep was bimodal), (2) the B=1 blocking host call
against host_pinned_max / se_fuse.  Run on the GPU box."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd"))
import torch  # noqa: F401,E402
from whenet_hip import _lib, weights as W  # noqa: E402


def rate(fn, n, secs=0.5):
    for _ in range(2):
        fn()
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < secs:
        fn()
        k += 1
    return k * n / (time.perf_counter() - t0)


def lat(fn, iters=500, drop=100):
    v = []
    for _ in range(iters):
        a = time.perf_counter()
        fn()
        v.append(time.perf_counter() - a)
    v = np.array(v[drop:]) * 1e6
    return float(np.median(v)), float(np.percentile(v, 99))


def main():
    blob = W.pack(W.synthetic(1234))
    rng = np.random.default_rng(0)
    big = rng.integers(0, 256, (1024, 224, 224, 3), dtype=np.uint8)
    h = _lib.Handle(blob, device=0, dtype=_lib.F16)
    cands = [(1, 0, 0, 0, 0), (2, 128, 1, 2, 256), (2, 128, 0, 2, 256), (3, 64, 1, 3, 256), (3, 64, 1, 1, 256), (2, 64, 1, 2, 256), (2, 256, 1, 2, 256), (3, 128, 1, 1, 256)]
    for N in (256, 512, 1024):
        crops = big[:N]
        res = {c: [] for c in cands}
        for trial in range(4):
            for c in cands:
                inflight, chunk, stage, depth, fmin = c
                h.set_option("inflight", inflight)
                h.set_option("fanout_min", fmin)
                if fmin:
                    h.set_option("fanout_chunk", chunk)
                    h.set_option("fanout_stage", stage)
                    h.set_option("fanout_depth", depth)
                res[c].append(rate(lambda: h.forward(crops, want_logits=False), N, 0.35))
        for c in cands:
            print(f"N={N:5d} inflight={c[0]} chunk={c[1]:3d} stage={c[2]} depth={c[3]} min={c[4]:3d}: " + " ".join(f"{v / 1e3:6.1f}k" for v in res[c]), flush=True)
    h.close()
    one = big[:1]
    for dt, name in ((_lib.F32, "f32"), (_lib.F16, "f16")):
        h = _lib.Handle(blob, device=0, dtype=dt)
        for pinned in (32, 0):
            for sf, tiny in ((1, 0), (0, 0), (2, 0)):
                h.set_option("host_pinned_max", pinned)
                h.set_option("se_fuse", sf)
                for wl in (True, False):
                    m, p99 = lat(lambda: h.forward(one, want_logits=wl))
                    print(f"B=1 {name} host call: pinned_max={pinned:2d} se_fuse={sf} logits={int(wl)}: median {m:6.1f} us  p99 {p99:6.1f} us", flush=True)
        h.set_option("se_fuse", 1)
        h.set_option("host_pinned_max", 32)
        for n in (2, 4, 8, 16, 32, 64):
            c = big[:n]
            a, _ = lat(lambda: h.forward(c, want_logits=True), 200, 50)
            h.set_option("host_pinned_max", 0)
            b, _ = lat(lambda: h.forward(c, want_logits=True), 200, 50)
            h.set_option("host_pinned_max", 32 if n <= 32 else 64)
            print(f"B={n} {name} host call: pinned {a:7.1f} us, pageable {b:7.1f} us", flush=True)
        h.close()
    import whenet
    for dd in ("f32", "f16"):
        with whenet.WHENet(dtype=dd) as m:
            med, p99 = lat(lambda: m.get_angle(one), 600)
            print(f"drop-in get_angle(uint8[1]) {dd}: median {med:.1f} us p99 {p99:.1f} us", flush=True)
            r = rate(lambda: m.get_angle(big[:512]), 512, 0.8)
            print(f"drop-in get_angle(uint8[512]) {dd}: {r:.0f} crops/s", flush=True)


if __name__ == "__main__":
    main()
