#!/usr/bin/env python3
"""Round-3 diagnostic: bench.py's RCCL branch (1 rank, strong scaling, 512 crops, forwards in flight) run repeatedly
as a CHILD of a process that itself holds two handles on the GPU -- the condition under which a replica engine's
ticket counters were zeroed late (engine.cpp ensure_capacity; tests/test_gpu_parity.py::
test_replica_engines_first_forward_under_load).  Prints the failures of bench.py's in-flight equality check."""
import os, sys, subprocess
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd")); sys.path.insert(0, ROOT)
import numpy as np
from whenet_hip import _lib, synth, weights as W
blob = W.pack(W.synthetic(1234))
hs = [_lib.Handle(blob, device=0, dtype=d) for d in (_lib.F16, _lib.F32)]
crops = synth.noise_crops(512, seed=3)
for h in hs:
    h.forward(crops)
print("parent holds 2 handles", flush=True)
env = dict(os.environ, WHENET_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
        "--no-latency", "--no-sweep", "--profile-iters", "2"]
import collections
res = collections.Counter()
cfgs = {"base": [], "inflight2": ["--inflight", "2"]}
for i in range(14):
    for name, extra in cfgs.items():
        r = subprocess.run(base + ["--strong", "--global-batch", "512"] + extra, env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            res[name] += 1
            print("FAIL", name, [ln[-300:] for ln in r.stderr.splitlines() if "differ" in ln or "Error" in ln][-1:], flush=True)
print("failures of 14:", dict(res), flush=True)
