#!/bin/bash
mkdir -p gpurun_out
for l in 3 4 5 6; do
timeout 300 python bench.py --lanes $l --steps 100 --warmup 10 --no-cpu-baseline --no-latency > gpurun_out/bench_lanes$l.txt 2>&1
python - <<PY
import json
for l in open("gpurun_out/bench_lanes$l.txt"):
    if l.startswith("{"):
        d=json.loads(l); print("lanes $l", round(d["value"]), "crops/s", round(d["ms_per_step"],3), "ms/step; chain", round(d["roofline"]["chain_us_per_step"]), "us")
PY
done
