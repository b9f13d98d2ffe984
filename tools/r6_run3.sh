cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6c
O=gpurun_out/r6c/ab_mb7_v3.txt; : > $O
run() { # dtype batch extra-args label
  python bench.py --dtype $1 --batch $2 --no-cpu-baseline --no-latency --no-sweep $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 b$2 [$4]: value %.0f  serial %.0f' % (d['value'], d.get('value_serial') or 0))" | tee -a $O
}
for r in 1 2; do
run f16 64 "--opt mb7=0" "mb7=0"
run f16 64 "--opt mb7=1" "mb7=1"
done
run f16 64 "--opt mb7=0 --inflight 4" "mb7=0 inflight4"
run f16 64 "--opt mb7=1 --inflight 4" "mb7=1 inflight4"
run f16 64 "--opt mb7=1 --lanes 1" "mb7=1 lanes1"
run f16 128 "--opt mb7=0" "mb7=0"
run f16 128 "--opt mb7=1" "mb7=1"
run f16 256 "--opt mb7=0" "mb7=0"
run f16 256 "--opt mb7=1" "mb7=1"
run f16 512 "--opt mb7=0" "mb7=0"
run f16 512 "--opt mb7=1" "mb7=1"
python tools/b1_probe.py 2>&1 | grep f16 | tee -a $O
