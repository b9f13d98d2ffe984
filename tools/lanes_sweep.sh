for q in 4 8; do for l in 3 4 6 8; do
GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --dtype f16 --batch 64 --lanes $l --steps 100 --warmup 10 --no-cpu-baseline --no-latency --profile-iters 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q lanes $l', round(d['value']), round(d['ms_per_step'],3))"
done; done
for l in 1 2 4; do GPU_MAX_HW_QUEUES=8 timeout 200 python bench.py --dtype f16 --batch 16 --lanes $l --steps 100 --warmup 10 --no-cpu-baseline --no-latency --profile-iters 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b16 lanes $l', round(d['value']), round(d['ms_per_step'],3))"; done
