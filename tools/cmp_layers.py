#!/usr/bin/env python3
"""compare two per-layer dumps: cmp_layers.py a.json b.json [filter]"""
import json, sys
a = json.load(open(sys.argv[1]))["launches"]; b = json.load(open(sys.argv[2]))["launches"]
flt = sys.argv[3] if len(sys.argv) > 3 else ""
print("total", round(sum(x["avg_us"] for x in a), 1), "->", round(sum(x["avg_us"] for x in b), 1))
kinds = {}
for x, y in zip(a, b):
    k = x["layer"].split("/")[-1]
    s = kinds.setdefault(k, [0.0, 0.0]); s[0] += x["avg_us"]; s[1] += y["avg_us"]
    if flt and flt in x["layer"]:
        print(f"  {x['layer']:12s} {x['avg_us']:8.1f} -> {y['avg_us']:8.1f}")
print({k: (round(v[0], 1), round(v[1], 1)) for k, v in kinds.items()})
