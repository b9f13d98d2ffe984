#!/bin/bash
# End-of-round evidence run (on the GPU box): full GPU test suite, default bench, rocprofv3 stats of
# the SAME bench command, PMC passes for the HBM traffic.  Outputs under gpurun_out/final; copy into profiles/rNN.
R=${GRAFT_REPO_ROOT:-$(pwd)}
RND=${RND:-r06}
mkdir -p $R/gpurun_out/final $R/profiles/$RND
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short > $R/gpurun_out/final/pytest_gpu.txt 2>&1; echo "pytest exit $?"; tail -3 $R/gpurun_out/final/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $R/gpurun_out/final/smoke.txt 2>&1; echo "smoke exit $?"; tail -3 $R/gpurun_out/final/smoke.txt
# PMC passes first: bench.py reads the per-kernel HBM traffic from profiles/$RND/pmc_traffic_<dtype>_b<B>.json, and only the set
# whose crops per launch equal the profiled launch's: 64 (the default line: 3 forwards in flight, one chain each) and 32
# (the serial schedule's two lanes)
rm -f $R/profiles/$RND/pmc_traffic_f16_b64.json
for LB in 64 32; do
  bash tools/pmc_round.sh f16 64 $LB > $R/gpurun_out/final/pmc_c$LB.log 2>&1; echo "pmc (crops per launch $LB) exit $?"
  python tools/pmc_summary.py gpurun_out/pmc_f16_b64_c${LB}_p profiles/$RND/pmc_traffic_f16_b64.json $LB gpurun_out/pmc_f16_b64_c${LB}_layers.json > $R/gpurun_out/final/pmc_f16_b64_c${LB}_by_kernel.txt 2>&1; echo "pmc summary exit $?"
done
cp profiles/$RND/pmc_traffic_f16_b64.json $R/gpurun_out/final/
# the parity configuration's chain (one chain of 64 crops)
rm -f $R/profiles/$RND/pmc_traffic_f32_b64.json
bash tools/pmc_round.sh f32 64 64 > $R/gpurun_out/final/pmc_f32_c64.log 2>&1; echo "pmc f32 exit $?"
python tools/pmc_summary.py gpurun_out/pmc_f32_b64_c64_p profiles/$RND/pmc_traffic_f32_b64.json 64 gpurun_out/pmc_f32_b64_c64_layers.json > $R/gpurun_out/final/pmc_f32_b64_c64_by_kernel.txt 2>&1; echo "pmc f32 summary exit $?"
cp profiles/$RND/pmc_traffic_f32_b64.json $R/gpurun_out/final/
timeout 600 python bench.py --dump-layers $R/gpurun_out/final/layers_default.json > $R/gpurun_out/final/bench_default.json 2> $R/gpurun_out/final/bench_default.err; echo "bench exit $?"
# the driver's own command line (20 timed steps: the region is repeated, the median is the value)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/final/bench_driver_args.json 2> /dev/null; echo "bench (driver args) exit $?"
timeout 300 python tools/f16_error_gpu.py 48 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > $R/gpurun_out/final/f16_error_gpu.txt
timeout 300 python bench.py --dtype f32 --no-cpu-baseline --no-latency --no-sweep --dump-layers $R/gpurun_out/final/layers_f32_b64.json > $R/gpurun_out/final/bench_f32_b64.json 2>/dev/null
timeout 300 python bench.py --batch 512 --no-cpu-baseline --no-latency --no-sweep > $R/gpurun_out/final/bench_f16_b512.json 2>/dev/null
timeout 300 python bench.py --batch 8 --no-cpu-baseline --no-latency --no-sweep > $R/gpurun_out/final/bench_f16_b8.json 2>/dev/null
timeout 300 python bench.py --batch 1 --no-cpu-baseline --no-latency --no-sweep > $R/gpurun_out/final/bench_f16_b1.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/final/rocprof_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/rocprof_stats -o bench -- python $R/bench.py --no-cpu-baseline --no-latency --no-serial --no-sweep > $R/gpurun_out/final/rocprof_bench.json 2>/dev/null; echo "rocprof exit $?"
rm -rf $R/gpurun_out/final/rocprof_stats_f32
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/rocprof_stats_f32 -o bench -- python $R/bench.py --dtype f32 --no-cpu-baseline --no-latency --no-serial --no-sweep > /dev/null 2>&1; echo "rocprof f32 exit $?"
rm -rf $R/gpurun_out/final/rocprof_stats_b512
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/rocprof_stats_b512 -o bench -- python $R/bench.py --batch 512 --inflight 1 --no-cpu-baseline --no-latency --no-serial --no-sweep > /dev/null 2>&1; echo "rocprof b512 exit $?"
cd $R && python tools/gpu_diag.py --quick > $R/gpurun_out/final/diag.txt 2>&1; echo "diag exit $?"
# ---- round 5 additions: the f32s configuration, counters at 256 crops per launch, the round's A/B and host-path probes
cd $R
timeout 300 python bench.py --dtype f32s --no-cpu-baseline --no-latency --no-sweep --dump-layers $R/gpurun_out/final/layers_f32s_b64.json > $R/gpurun_out/final/bench_f32s_b64.json 2>/dev/null; echo "bench f32s exit $?"
rm -f $R/profiles/$RND/pmc_traffic_f32s_b64.json
bash tools/pmc_round.sh f32s 64 64 > $R/gpurun_out/final/pmc_f32s_c64.log 2>&1; echo "pmc f32s exit $?"
python tools/pmc_summary.py gpurun_out/pmc_f32s_b64_c64_p profiles/$RND/pmc_traffic_f32s_b64.json 64 gpurun_out/pmc_f32s_b64_c64_layers.json > $R/gpurun_out/final/pmc_f32s_b64_c64_by_kernel.txt 2>&1; echo "pmc f32s summary exit $?"
rm -f $R/profiles/$RND/pmc_traffic_f16_b512.json
bash tools/pmc_round.sh f16 512 256 > $R/gpurun_out/final/pmc_c256.log 2>&1; echo "pmc (256 crops per launch) exit $?"
python tools/pmc_summary.py gpurun_out/pmc_f16_b512_c256_p profiles/$RND/pmc_traffic_f16_b512.json 256 > $R/gpurun_out/final/pmc_f16_b512_c256_by_kernel.txt 2>&1; echo "pmc 256 summary exit $?"
timeout 300 python tools/staged_ab.py f16 f32s 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > $R/gpurun_out/final/staged_splitk_ab.txt; echo "staged ab exit $?"
timeout 300 python tools/numa_probe.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > $R/gpurun_out/final/host_path_numa_probe.txt; echo "numa probe exit $?"
tools/probes/fetch_pattern_probe > $R/gpurun_out/final/fetch_pattern_probe.txt 2>&1; echo "fetch probe exit $?"
tools/probes/mfma_denorm_probe > $R/gpurun_out/final/mfma_denorm_probe.txt 2>&1; echo "denorm probe exit $?"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/final/rocprof_stats_f32s
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/rocprof_stats_f32s -o bench -- python $R/bench.py --dtype f32s --no-cpu-baseline --no-latency --no-serial --no-sweep > /dev/null 2>&1; echo "rocprof f32s exit $?"
# ---- round 6 additions: the one-launch 7 x 7 block kernel (probe timeline, A/B), XCD placement A/B, host paths
cd $R
tools/probes/mb7_probe > $R/gpurun_out/final/mb7_probe_timeline.txt 2>&1; echo "mb7 probe exit $?"
timeout 600 bash tools/ab_opts.sh f16 64 "mb7=0" "mb7=1" "xcd_map=0" "xcd_map=7" > $R/gpurun_out/final/ab_mb7_xcd_f16_b64.txt 2>&1; echo "ab b64 exit $?"
timeout 600 bash tools/ab_opts.sh f16 512 "mb7=0" "mb7=1" "xcd_map=0" "xcd_map=7" > $R/gpurun_out/final/ab_mb7_xcd_f16_b512.txt 2>&1; echo "ab b512 exit $?"
timeout 600 bash tools/ab_opts.sh f32s 64 "xcd_map=0" "xcd_map=7" > $R/gpurun_out/final/ab_xcd_f32s_b64.txt 2>&1; echo "ab f32s exit $?"
timeout 600 python tools/latency_ab.py "mb7=0" "mb7=1" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > $R/gpurun_out/final/latency_b1_mb7.txt; echo "latency ab exit $?"
timeout 900 python tools/host_path_probe.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > $R/gpurun_out/final/host_path_probe.txt; echo "host path probe exit $?"
# PMC of the one-launch blocks (f16, 64 crops per launch, option mb7=1)
PMC_EXTRA="--opt mb7=1" bash tools/pmc_round.sh f16 64 64 > $R/gpurun_out/final/pmc_mb7_c64.log 2>&1; echo "pmc mb7 exit $?"
cp -r gpurun_out/pmc_f16_b64_c64_layers.json gpurun_out/pmc_f16_b64_c64_mb7_layers.json
python tools/pmc_summary.py gpurun_out/pmc_f16_b64_c64_p /tmp/pmc_traffic_mb7.json 64 gpurun_out/pmc_f16_b64_c64_mb7_layers.json > $R/gpurun_out/final/pmc_f16_b64_c64_mb7_by_kernel.txt 2>&1; echo "pmc mb7 summary exit $?"
