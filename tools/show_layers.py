#!/usr/bin/env python3
"""Pretty-print a per-launch profile written by `bench.py --dump-layers`."""
import json
import sys

d = json.load(open(sys.argv[1]))
tot = sum(s["avg_us"] for s in d["launches"])
print(f"batch {d['batch']} dtype {d['dtype']}: {len(d['launches'])} launches, {tot:.1f} us/step")
print(f"{'layer':14s}{'kind':7s}{'us':>9s}{'GB/s':>9s}{'TF/s':>8s}  ideal_us@6.3TB/s")
for s in d["launches"]:
    us = s["avg_us"]
    print(f"{s['layer']:14s}{s['kind']:7s}{us:9.2f}{s['alg_bytes'] / us / 1e3:9.1f}{s['alg_flops'] / us / 1e6:8.2f}"
          f"  {s['alg_bytes'] / 6.3e6:8.2f}")
kinds = {}
for s in d["launches"]:
    k = kinds.setdefault(s["kind"], [0.0, 0.0])
    k[0] += s["avg_us"]
    k[1] += s["alg_bytes"]
for k, (us, by) in kinds.items():
    print(f"  {k:6s} {us:9.1f} us  {by / us / 1e3:8.1f} GB/s   ideal {by / 6.3e6:8.1f} us")
