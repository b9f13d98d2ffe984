for M in 2 3 4; do python bench.py --inflight $M --no-cpu-baseline --no-latency --no-sweep --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight', d['config']['forwards_in_flight'], round(d['value']), round(d['value_serial'] or 0))"; done
for B in 128 256; do python bench.py --batch $B --no-cpu-baseline --no-latency --no-sweep --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch', d['config']['batch_per_gpu'], round(d['value']), round(d['value_serial'] or 0))"; done
