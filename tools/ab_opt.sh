#!/bin/bash
# A/B of engine options under the default bench line: [AB_BENCH_ARGS="--dtype f32"] tools/ab_opt.sh "se_fuse=2" "se_fuse=1" "se_fuse=0 fold12=0" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for v in "$@"; do
  O=""; for kv in $v; do O="$O --opt $kv"; done
  timeout 300 python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-latency --no-sweep $AB_BENCH_ARGS $O 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-36s value %9.0f serial %9.0f chain %7.1f us' % ('$v', d['value'], d['value_serial'], d['roofline']['chain_us_per_step']))"
done
