#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r05
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short > $R/gpurun_out/r05/pytest_gpu_call3.txt 2>&1; echo "pytest exit $?"; grep -E "passed|failed|error" $R/gpurun_out/r05/pytest_gpu_call3.txt | tail -3
( time timeout 600 python bench.py > $R/gpurun_out/r05/bench_default_call3.json 2> $R/gpurun_out/r05/bench_default_call3.err ) 2>&1 | grep real; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_default_call3.json').read().strip().split('\n')[-1])
print({k:d[k] for k in ('metric','value','unit','ms_per_step','dtype')})
print('roofline', d.get('roofline'))
print('cpu_baseline', d.get('cpu_baseline'))
print('check', d.get('check'))
print('dropin', d.get('dropin'))
print('pcie', d.get('pcie_inclusive'))
print('latency_b1', d.get('latency_b1'))
sw=d.get('sweep',{})
for k,v in sw.items():
    if isinstance(v,dict): print(k, round(v['value']), round(v['value_serial']), v['dtype'])
PY
