#!/bin/bash
# per-launch profiles of the default batch-64 f16 bench under several engine options: tools/ab_layers.sh "front_impl=0" "front_impl=2" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
i=0
for OPT in "$@"; do
  i=$((i+1))
  ARGS=""; LIBV=""; for kv in $OPT; do case $kv in lib=*) LIBV=$R/headposeestimation-whenet_amd/lib/variants/lib_${kv#lib=}.so;; *) ARGS="$ARGS --opt $kv";; esac; done
  if [ -n "$LIBV" ]; then export WHENET_HIP_LIB=$LIBV; else unset WHENET_HIP_LIB; fi
  timeout 300 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-latency --no-sweep $ARGS --dump-layers $R/gpurun_out/ab_layers_$i.json > $R/gpurun_out/ab_bench_$i.json 2>/dev/null
  python - "$R/gpurun_out/ab_bench_$i.json" "$OPT" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:40s} value {d['value']:9.0f} serial {d['value_serial']:9.0f} chain {d['roofline']['chain_us_per_step']:7.1f} us  check {d['check']['max_abs_deg_vs_f64_oracle']:.3f}")
PY
done
python - $R/gpurun_out $i <<'PY'
import json,sys
n=int(sys.argv[2]); L=[json.load(open(f"{sys.argv[1]}/ab_layers_{k}.json"))['launches'] for k in range(1,n+1)]
for rows in zip(*L):
    if rows[0]['kind']=='calib': continue
    print(f"{rows[0]['layer']:14s}"+"".join(f" {r['avg_us']:8.2f}" for r in rows)+"   "+rows[0]['kernel'][:50])
PY
