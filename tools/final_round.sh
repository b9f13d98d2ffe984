#!/bin/bash
# End-of-round evidence run (on the GPU box): full GPU test suite, default bench, rocprofv3 stats of
# the SAME bench command, PMC passes for the HBM traffic.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/final
timeout 900 python -m pytest tests -m gpu -q --tb=short > $R/gpurun_out/final/pytest_gpu.txt 2>&1; echo "pytest exit $?"; tail -3 $R/gpurun_out/final/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $R/gpurun_out/final/smoke.txt 2>&1; echo "smoke exit $?"; tail -3 $R/gpurun_out/final/smoke.txt
# PMC passes first: bench.py reads the per-kernel HBM traffic from profiles/r03/pmc_traffic_<dtype>_b<B>.json
cd $R && bash tools/pmc_round.sh f16 64 > $R/gpurun_out/final/pmc.log 2>&1; echo "pmc exit $?"
python tools/pmc_summary.py gpurun_out/pmc_f16_b64_p profiles/r03/pmc_traffic_f16_b64.json > $R/gpurun_out/final/pmc_f16_b64_by_kernel.txt 2>&1; echo "pmc summary exit $?"
cp profiles/r03/pmc_traffic_f16_b64.json $R/gpurun_out/final/
timeout 600 python bench.py --dump-layers $R/gpurun_out/final/layers_default.json > $R/gpurun_out/final/bench_default.json 2> $R/gpurun_out/final/bench_default.err; echo "bench exit $?"
# the driver's own command line (20 timed steps: the region is repeated, the median is the value)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/final/bench_driver_args.json 2> /dev/null; echo "bench (driver args) exit $?"
python tools/f16_error_gpu.py 48 > $R/gpurun_out/final/f16_error_gpu.txt 2>&1
timeout 300 python bench.py --dtype f32 --no-cpu-baseline --no-latency --no-sweep > $R/gpurun_out/final/bench_f32_b64.json 2>/dev/null
timeout 300 python bench.py --batch 512 --no-cpu-baseline --no-latency --no-sweep > $R/gpurun_out/final/bench_f16_b512.json 2>/dev/null
timeout 300 python bench.py --batch 8 --no-cpu-baseline --no-latency --no-sweep > $R/gpurun_out/final/bench_f16_b8.json 2>/dev/null
timeout 300 python bench.py --batch 1 --no-cpu-baseline --no-latency --no-sweep > $R/gpurun_out/final/bench_f16_b1.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/final/rocprof_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/rocprof_stats -o bench -- python $R/bench.py --no-cpu-baseline --no-latency --no-serial --no-sweep > $R/gpurun_out/final/rocprof_bench.json 2>/dev/null; echo "rocprof exit $?"
cd $R && python tools/gpu_diag.py --quick > $R/gpurun_out/final/diag.txt 2>&1; echo "diag exit $?"
