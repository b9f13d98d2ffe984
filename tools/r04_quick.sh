#!/bin/bash
# quick GPU check: [tests] + one bench line (no CPU baseline) + per-layer dump.  usage: tools/r04_quick.sh <tag> [pytest -k expr | "all" | "none"] [bench args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-q}; K=${2:-all}; shift 2
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
if [ "$K" = "all" ]; then
  timeout 900 python -m pytest tests -m gpu -q --tb=short -x > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?"
elif [ "$K" != "none" ]; then
  timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "$K" > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?"
fi
[ -f $O/pytest_gpu.txt ] && grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest_gpu.txt | tail -25
timeout 600 python bench.py --no-cpu-baseline --dump-layers $O/layers.json "$@" > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; tail -c 600 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:(round(d[k],4) if isinstance(d[k],float) else d[k]) for k in ("value","value_serial","ms_per_step","ms_per_step_serial")})
r=d["roofline"]; print({k:r[k] for k in ("kernel","avg_launch_us","frac","crops_per_launch","chain_us_per_step")})
print("check", d.get("check"))
sw=d.get("sweep") or {}
print("sweep", {k:(round(v["value"]),round(v["value_serial"])) for k,v in sw.items() if isinstance(v,dict)})
print("lat", d.get("latency_b1")); print("pcie", d.get("pcie_inclusive"))
PY
python tools/show_layers.py $O/layers.json | tail -30
