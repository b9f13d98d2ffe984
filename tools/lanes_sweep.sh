for l in 1 2 3 4; do timeout 200 python bench.py --dtype f16 --batch 64 --lanes $l --steps 100 --warmup 10 --no-cpu-baseline --no-latency --profile-iters 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $l', round(d['value']), round(d['ms_per_step'],3))"; done
