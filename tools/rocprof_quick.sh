#!/bin/bash
# rocprofv3 kernel stats of a short bench run: tools/rocprof_quick.sh <tag> <bench args...>
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/rp_$TAG
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rp_$TAG -o s -- python $R/bench.py --no-cpu-baseline --no-latency --no-sweep --no-serial --steps 100 --warmup 10 "$@" > /dev/null 2>&1
python3 - $R/gpurun_out/rp_$TAG <<'PY'
import csv,glob,sys,re,subprocess
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:40]:
    n=r['Name']
    n=re.sub(r'void whenet::\(anonymous namespace\)::','',n); n=re.sub(r'_ZN6whenet12_GLOBAL__N_1\d+','',n)
    print(f"{n[:78]:78s} calls={r['Calls']:>6} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={float(r['Percentage']):5.1f}")
PY
