#!/bin/bash
# A/B of an environment variable: [AB_BENCH_ARGS="--dtype f32"] tools/ab_env.sh VAR v1 v2 ...  -> value / serial / chain and the per-layer table
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
VAR=$1; shift; i=0
for v in "$@"; do
  i=$((i+1))
  env $VAR=$v timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-latency --no-sweep $AB_BENCH_ARGS --dump-layers gpurun_out/abe_layers_$i.json > gpurun_out/abe_bench_$i.json 2>/dev/null
  python - gpurun_out/abe_bench_$i.json "$VAR=$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:30s} value {d['value']:9.0f} serial {d['value_serial']:9.0f} chain {d['roofline']['chain_us_per_step']:7.1f} us")
PY
done
python - gpurun_out $i <<'PY'
import json,sys
n=int(sys.argv[2]); L=[json.load(open(f"{sys.argv[1]}/abe_layers_{k}.json"))['launches'] for k in range(1,n+1)]
for rows in zip(*L):
    if rows[0]['kind']!='pw': continue
    print(f"{rows[0]['layer']:14s}"+"".join(f" {r['avg_us']:8.2f}" for r in rows)+"   "+rows[0]['kernel'][:50])
PY
