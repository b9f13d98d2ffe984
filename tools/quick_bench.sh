#!/bin/bash
# GPU-box quick check: parity subset + default-shape benches (args: extra bench options)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x > gpurun_out/pytest_quick.txt 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_quick.txt
for cfg in "f16 64" "f32 64" "f16 512" "f16 1" "f16 8"; do set -- $cfg
timeout 300 python bench.py --dtype $1 --batch $2 --steps 100 --warmup 10 --no-cpu-baseline --no-latency $QB_OPTS --dump-layers gpurun_out/layers_q_$1_b$2.json > gpurun_out/bench_q_$1_b$2.txt 2>&1; echo "bench $cfg exit $?"
python - <<PY
import json
for l in open("gpurun_out/bench_q_$1_b$2.txt"):
    if l.startswith("{"):
        d=json.loads(l); print("$cfg", round(d["value"]), "crops/s", round(d["ms_per_step"],3), "ms/step; chain", round(d["roofline"]["chain_us_per_step"]), "us")
PY
done
python tools/show_layers.py gpurun_out/layers_q_f16_b64.json > gpurun_out/layers_q_f16_b64.txt 2>&1
