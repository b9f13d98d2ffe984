mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -x > gpurun_out/pytest.txt 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest.txt
for cfg in "f16 64 4" "f32 64 4" "f16 512 4" "f16 1 1" "f32 1 1"; do set -- $cfg
timeout 300 python bench.py --dtype $1 --batch $2 --lanes $3 --steps 50 --warmup 5 --no-cpu-baseline --no-latency --dump-layers gpurun_out/layers_$1_b$2.json > gpurun_out/bench_$1_b$2_l$3.txt 2>&1; echo "bench $cfg exit $?"
python - <<PY
import json
for l in open("gpurun_out/bench_$1_b$2_l$3.txt"):
    if l.startswith("{"):
        d=json.loads(l); print("$cfg", round(d["value"]), "crops/s", round(d["ms_per_step"],3), "ms/step; chain", round(d["roofline"]["chain_us_per_step"]), "us")
PY
done
