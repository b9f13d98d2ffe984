#!/bin/bash
# round 4, first GPU call: the suite, the f16 error distribution against the oracle fixtures, one bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -x > $O/pytest_gpu.txt 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_gpu.txt
timeout 300 python tools/f16_error_gpu.py 48 > $O/f16_error_gpu.txt 2>&1; echo "f16 error exit $?"; cat $O/f16_error_gpu.txt | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
timeout 600 python bench.py --dump-layers $O/layers_default.json > $O/bench_default.json 2> $O/bench_default.err; echo "bench exit $?"; tail -c 1500 $O/bench_default.err
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","value_serial","ms_per_step","host_enqueue_ms_per_step")})
r=d["roofline"]; print({k:r[k] for k in ("kernel","avg_launch_us","frac","crops_per_launch","traffic","traffic_source","alg_bytes_per_launch")}); print(r["valu"]["fused_front_kernels"])
print(d.get("check")); print(d.get("cpu_baseline")); print(d.get("sweep")); print(d.get("latency_b1")); print(d.get("pcie_inclusive")); print(d["config"]["cpu_affinity"])
PY
