#!/bin/bash
# Re-run the parts of the evidence set that a late change touches: GPU test suite, smoke, the two bench lines.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/final
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short > $R/gpurun_out/final/pytest_gpu.txt 2>&1; echo "pytest exit $?"; grep -E "passed|failed" $R/gpurun_out/final/pytest_gpu.txt | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $R/gpurun_out/final/smoke.txt 2>&1; echo "smoke exit $?"; tail -2 $R/gpurun_out/final/smoke.txt
timeout 600 python bench.py --dump-layers $R/gpurun_out/final/layers_default.json > $R/gpurun_out/final/bench_default.json 2> $R/gpurun_out/final/bench_default.err; echo "bench exit $?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/final/bench_driver_args.json 2> /dev/null; echo "bench (driver args) exit $?"
python - <<'PY'
import json
for f in ("bench_default.json","bench_driver_args.json"):
    d=json.loads(open("gpurun_out/final/"+f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d.get("value_serial") or 0), d["roofline"]["kernel"], round(d["roofline"]["frac"],3), d.get("latency_b1"), d.get("dropin"))
PY
