cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6e
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6e/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r6e/pytest_gpu.txt
timeout 600 python bench.py > gpurun_out/r6e/bench_default.json 2> gpurun_out/r6e/bench_default.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r6e/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','unit','ms_per_step','dtype','vs_baseline')})
print('value_serial', d.get('value_serial'), 'parity_value', d.get('parity_value'), d.get('parity_dtype'))
print('roofline', {k:d['roofline'].get(k) for k in ('kernel','layer','frac','achieved','traffic')})
print('dropin', d.get('dropin')); print('pcie', d.get('pcie_inclusive')); print('latency_b1', d.get('latency_b1')); print('sweep', {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d.get('sweep',{}).items()})
P
