#!/bin/bash
# how many chains in flight, and how they are arranged (engines x lanes): value of the default bench line  [B=64 CS_ARGS="--dtype f32"]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for cfg in "3 0" "2 2" "3 2" "4 0" "2 0" "1 2" "1 3" "4 2"; do
  set -- $cfg
  L=""; [ "$2" != "0" ] && L="--lanes $2"
  timeout 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-latency --no-sweep --no-serial --inflight $1 $L --batch ${B:-64} $CS_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('inflight $1 lanes $2 batch ${B:-64}: value %.0f crops/s  ms/step %.4f' % (d['value'], d['ms_per_step']))"
done
