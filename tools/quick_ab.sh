#!/bin/bash
# A/B helper: GPU tests (quiet) + B=64 and B=512 bench lines with per-layer dumps under gpurun_out/$1
T=${1:-ab}
mkdir -p gpurun_out/$T
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 100 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --dump-layers gpurun_out/$T/layers64.json 2>/dev/null | tail -1 > gpurun_out/$T/bench64.json
timeout 100 python bench.py --batch 512 --steps 30 --warmup 5 --no-cpu-baseline --no-latency --dump-layers gpurun_out/$T/layers512.json 2>/dev/null | tail -1 > gpurun_out/$T/bench512.json
python - <<PY
import json
for f in ("bench64","bench512"):
    d=json.loads(open("gpurun_out/$T/%s.json"%f).read())
    print(f, round(d["value"]), "serial", round(d["serial_schedule"]["value"]), "b1f32", (d.get("latency_b1") or {}).get("median_us"))
PY
