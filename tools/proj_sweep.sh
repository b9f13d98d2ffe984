#!/bin/bash
# GPU-box sweep: fused SE+project from block index F (0 = never) at the default bench shape.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "mbconv or info or batch_invariance or golden" > gpurun_out/pytest_proj.txt 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/pytest_proj.txt
for cfg in "f16 64 0" "f16 64 1" "f16 64 4" "f16 64 7" "f16 64 13" "f32 64 0" "f32 64 1" "f16 512 0" "f16 512 1" "f16 1 0" "f16 1 1"; do set -- $cfg
timeout 300 python bench.py --dtype $1 --batch $2 --opt fuse_project=$3 --steps 100 --warmup 10 --no-cpu-baseline --no-latency --dump-layers gpurun_out/layers_proj_$1_b$2_f$3.json > gpurun_out/bench_proj_$1_b$2_f$3.txt 2>&1; echo "bench $cfg exit $?"
python - <<PY
import json
for l in open("gpurun_out/bench_proj_$1_b$2_f$3.txt"):
    if l.startswith("{"):
        d=json.loads(l); print("$cfg", round(d["value"]), "crops/s", round(d["ms_per_step"],3), "ms/step; chain", round(d["roofline"]["chain_us_per_step"]), "us")
PY
done
python tools/show_layers.py gpurun_out/layers_proj_f16_b64_f0.json > gpurun_out/layers_proj_f0.txt 2>&1
python tools/show_layers.py gpurun_out/layers_proj_f16_b64_f1.json > gpurun_out/layers_proj_f1.txt 2>&1
