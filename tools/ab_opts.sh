#!/bin/bash
# A/B of engine options through bench.py on the GPU box: one line per option set (value = 3 forwards in flight, serial = one at a time).
# Usage: bash tools/ab_opts.sh <dtype> <batch> "<opt set 1>" "<opt set 2>" ...   an opt set = space-separated KEY=VALUE (or "-" for none)
DT=${1:-f16}; B=${2:-64}; shift 2
for round in 1 2; do
for SET in "$@"; do
  OPTS=""
  if [ "$SET" != "-" ]; then for kv in $SET; do OPTS="$OPTS --opt $kv"; done; fi
  python bench.py --dtype $DT --batch $B --no-cpu-baseline --no-latency --no-sweep $OPTS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$DT b$B [$SET] round $round: value %.0f  serial %.0f' % (d['value'], d.get('value_serial') or 0))"
done
done
