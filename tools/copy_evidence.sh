#!/bin/bash
# gpurun_out/final (tools/final_round.sh on the GPU box) -> profiles/rNN; prints the numbers the docs quote
R=$(cd "$(dirname "$0")/.." && pwd); RND=${RND:-r06}; F=$R/gpurun_out/final; P=$R/profiles/$RND
cp $F/bench_default.json $F/bench_driver_args.json $F/bench_f16_b1.json $F/bench_f16_b8.json $F/bench_f16_b512.json $F/bench_f32_b64.json \
   $F/layers_default.json $F/layers_f32_b64.json $F/smoke.txt $F/diag.txt $F/f16_error_gpu.txt $F/rocprof_bench.json \
   $F/pmc_f16_b64_c64_by_kernel.txt $F/pmc_f16_b64_c32_by_kernel.txt $F/pmc_f32_b64_c64_by_kernel.txt $F/pmc_traffic_f16_b64.json $F/pmc_traffic_f32_b64.json $P/
mkdir -p $P
cp $F/bench_f32s_b64.json $F/layers_f32s_b64.json $F/pmc_f32s_b64_c64_by_kernel.txt $F/pmc_f16_b512_c256_by_kernel.txt $F/staged_splitk_ab.txt \
   $F/host_path_numa_probe.txt $F/fetch_pattern_probe.txt $F/mfma_denorm_probe.txt $P/ 2>/dev/null
cp $F/mb7_probe_timeline.txt $F/ab_mb7_xcd_f16_b64.txt $F/ab_mb7_xcd_f16_b512.txt $F/ab_xcd_f32s_b64.txt $F/latency_b1_mb7.txt $F/host_path_probe.txt \
   $F/pmc_f16_b64_c64_mb7_by_kernel.txt $P/ 2>/dev/null
cp $F/rocprof_stats_f32s/bench_kernel_stats.csv $P/rocprofv3_kernel_stats_f32s_b64.csv 2>/dev/null
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $F/pytest_gpu.txt > $P/pytest_gpu.txt
cat $F/pmc_c64.log $F/pmc_c32.log $F/pmc_f32_c64.log $F/pmc_f32s_c64.log $F/pmc_c256.log | grep "^pmc pass" > $P/pmc.log
cp $F/rocprof_stats/bench_kernel_stats.csv $P/rocprofv3_kernel_stats_bench_default.csv
cp $F/rocprof_stats/bench_domain_stats.csv $P/rocprofv3_domain_stats_bench_default.csv
cp $F/rocprof_stats_b512/bench_kernel_stats.csv $P/rocprofv3_kernel_stats_b512_f16_serial.csv
cp $F/rocprof_stats_f32/bench_kernel_stats.csv $P/rocprofv3_kernel_stats_f32_b64.csv
tail -1 $P/pytest_gpu.txt
python3 - $P <<'PY'
import json,sys,collections
P=sys.argv[1]
def last(f): return json.loads(open(P+'/'+f).read().strip().splitlines()[-1])
for f in ["bench_default.json","bench_driver_args.json","bench_f32_b64.json","bench_f16_b512.json","bench_f16_b8.json","bench_f16_b1.json","rocprof_bench.json"]:
    d=last(f); r=d["roofline"]
    print(f, "value %.0f serial %.0f ms %.4f / %.4f"%(d["value"],d.get("value_serial") or 0,d["ms_per_step"],d.get("ms_per_step_serial") or 0), "| dom", r["kernel"][:44], "us %.2f frac %.3f traffic %s"%(r["avg_launch_us"],r["frac"],r.get("traffic")))
d=last("bench_default.json")
print("   sweep",{k:(round(v["value"]),round(v["value_serial"])) for k,v in d["sweep"].items() if isinstance(v,dict)})
print("   latency",d["latency_b1"]); print("   cpu",d["cpu_baseline"]["value"],d["cpu_baseline"]["single_process"]["value"]); print("   pcie",d["pcie_inclusive"])
print("   path", d["path_fraction"]["vs_2kernel_fusion_bound"], d["path_fraction"]["vs_layer_granular_bound"], "valu", d["roofline"]["valu"]["fused_front_kernels"])
print("   check", d["check"]["max_abs_deg_vs_f64_oracle"], d["check"]["argmax_flips"], "pmc", d["roofline"].get("pmc"))
print("   frame", {k:(round(v["sync_median_us"]),round(v["pipelined_frames_per_s"])) for k,v in d["frame_pipeline"].items() if isinstance(v,dict)}, "yolo", d["yolo_postprocess"]["gpu_median_us"])
for f in ("layers_default.json","layers_f32_b64.json"):
    L=json.load(open(P+'/'+f))['launches']; c=collections.Counter()
    for x in L: c[x['kind']]+=x['avg_us']
    print(f,{k:round(v,1) for k,v in c.items()}, round(sum(v for k,v in c.items() if k!='calib'),1), len(L)-1)
PY
