#!/bin/bash
# A/B of environment settings: [AB_BENCH_ARGS="--dtype f32"] tools/ab_env2.sh "A=1 B=2" "A=3" ...  -> value / serial / chain per setting, and the per-layer table
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
i=0
for v in "$@"; do
  i=$((i+1))
  env $v timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-latency --no-sweep $AB_BENCH_ARGS --dump-layers gpurun_out/abe_layers_$i.json > gpurun_out/abe_bench_$i.json 2>gpurun_out/abe_err_$i.txt
  python - gpurun_out/abe_bench_$i.json "$v" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:44s} value {d['value']:9.0f} serial {d['value_serial']:9.0f} chain {d['roofline']['chain_us_per_step']:7.1f} us  max err {d['check']['max_abs_deg_vs_f64_oracle']:.2e}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
python - gpurun_out $i ${AB_KIND:-front} <<'PY'
import json,sys
n=int(sys.argv[2]); L=[]
for k in range(1,n+1):
    try: L.append(json.load(open(f"{sys.argv[1]}/abe_layers_{k}.json"))['launches'])
    except Exception: pass
for rows in zip(*L):
    if rows[0]['kind'] not in sys.argv[3].split(','): continue
    print(f"{rows[0]['layer']:14s}"+"".join(f" {r['avg_us']:8.2f}" for r in rows))
PY
