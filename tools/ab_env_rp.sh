#!/bin/bash
# A/B of an environment variable under rocprofv3 (kernel stats of the default 3-in-flight bench): tools/ab_env_rp.sh VAR PATTERN v1 v2 ..
R=${GRAFT_REPO_ROOT:-$(pwd)}; VAR=$1; PAT=$2; shift 2
for v in "$@"; do
  echo "== $VAR=$v"
  env $VAR=$v bash $R/tools/rocprof_quick.sh ab_$v 2>&1 | grep -E "$PAT"
  cd $R && env $VAR=$v timeout 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-latency --no-sweep 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   value %.0f serial %.0f' % (d['value'], d['value_serial']))"
done
