#!/usr/bin/env python3
"""Is the box-to-box / run-to-run spread of the host-pointer paths a NUMA effect?  Same measurements with the process
unbound, bound to the GPU's NUMA node, and bound to another node (first-touch memory follows the CPUs)."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd"))
import torch
import ctypes
_libc = ctypes.CDLL(None)
def getcpu():
    try:
        return _libc.sched_getcpu()
    except Exception:
        return -1
from whenet_hip import _lib, weights as W
from whenet_hip.shard import gpu_numa_cpus, gpu_pci_bus_id, _parse_cpulist

def rate(fn, n, secs=0.4):
    for _ in range(2): fn()
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < secs:
        fn(); k += 1
    return k * n / (time.perf_counter() - t0)

bus = gpu_pci_bus_id(0)
node, cpus = gpu_numa_cpus(bus)
allc = sorted(os.sched_getaffinity(0))
print("GPU", bus, "numa node", node, "cpus", len(cpus), "| process affinity", len(allc), "cpus; running on cpu", getcpu(), flush=True)
nodes = {}
for d in sorted(os.listdir("/sys/devices/system/node")):
    if d.startswith("node"):
        try:
            nodes[int(d[4:])] = _parse_cpulist(open(f"/sys/devices/system/node/{d}/cpulist").read())
        except OSError:
            pass
print("nodes:", {k: len(v) for k, v in nodes.items()}, flush=True)
blob = W.pack(W.synthetic(1234))
configs = [("unbound", allc)]
if node >= 0 and cpus:
    configs.append((f"gpu node {node}", [c for c in cpus if c in allc]))
others = [k for k in nodes if k != node and any(c in allc for c in nodes[k])]
if others:
    configs.append((f"far node {others[-1]}", [c for c in nodes[others[-1]] if c in allc]))
for name, cs in configs:
    if not cs:
        continue
    os.sched_setaffinity(0, cs)
    time.sleep(0.05)
    rng = np.random.default_rng(0)
    big = rng.integers(0, 256, (512, 224, 224, 3), dtype=np.uint8)       # first touch on the bound CPUs
    h = _lib.Handle(blob, device=0, dtype=_lib.F16)                      # pinned staging allocated while bound
    res = {}
    h.set_option("fanout_min", 0)
    res["B=64 blocking"] = rate(lambda: h.forward(big[:64], want_logits=False), 64)
    res["N=512 one forward"] = rate(lambda: h.forward(big, want_logits=False), 512)
    h.set_option("fanout_min", 256)
    for infl in (2, 3, 4):
        h.set_option("inflight", infl)
        for stage in (0, 1):
            for chunk in (64, 128):
                h.set_option("fanout_stage", stage); h.set_option("fanout_chunk", chunk)
                res[f"N=512 fanout {infl}x{chunk} stage{stage}"] = rate(lambda: h.forward(big, want_logits=False), 512)
    h.set_option("inflight", 3)
    pend = []; t0 = time.perf_counter(); done = 0
    for i in range(90):
        if len(pend) == 3:
            h.collect(pend.pop(0), 64); done += 1
        pend.append(h.submit(big[:64]))
    while pend:
        h.collect(pend.pop(0), 64); done += 1
    res["submit/collect 3 in flight B=64"] = done * 64 / (time.perf_counter() - t0)
    h.close()
    print(f"[{name}: {len(cs)} cpus, on cpu {getcpu()}] " + "  ".join(f"{k}: {v / 1e3:.1f}k" for k, v in res.items()), flush=True)
    os.sched_setaffinity(0, allc)
