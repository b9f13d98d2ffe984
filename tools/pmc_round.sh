#!/bin/bash
# PMC counter passes (separate passes; rocprofv3 --pmc must not be combined with tracing other
# than --kernel-trace).  Usage: bash tools/pmc_round.sh <dtype> <batch> [crops_per_launch]
R=${GRAFT_REPO_ROOT:-$(pwd)}
DT=${1:-f16}; B=${2:-64}
cd /tmp && export TMPDIR=/tmp
# The counters are device-wide between a kernel's start and end: with several sub-batch chains (lanes) running side
# by side every kernel's figures would include the other chains' traffic (round 3 found the round-2 / first round-3
# tables inflated by exactly the number of lanes).  So the passes run ONE chain alone, --lanes 1, one forward at a time,
# of as many crops as a launch of the schedule in question processes:
#   crops_per_launch = B      the default bench line (3 forwards in flight: each forward is ONE chain of the batch) -- default
#   crops_per_launch = B / 2  the serial schedule (two lanes per forward)
# The output directories carry the crops per launch (pmc_<dtype>_b<B>_c<LB>_p<i>); tools/pmc_summary.py stores it in the
# traffic JSON and bench.py only prints a traffic figure whose crops per launch equal the profiled launch's.
LB=${3:-$B}
# (--dump-layers: the chain's launch list of THIS command, which tools/pmc_summary.py uses to name every dispatch's layer)
CMD="python $R/bench.py --dtype $DT --batch $LB --lanes 1 --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-serial --no-sweep --no-repeat --profile-iters 1 --no-graph --opt concurrent=1 --dump-layers $R/gpurun_out/pmc_${DT}_b${B}_c${LB}_layers.json $PMC_EXTRA"
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_${DT}_b${B}_c${LB}_p$i
  timeout 400 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $R/gpurun_out/pmc_${DT}_b${B}_c${LB}_p$i -o p -- $CMD > $R/gpurun_out/pmc_${DT}_b${B}_c${LB}_p$i.log 2>&1
  rc=$?
  # (exit 1 with a complete counter_collection.csv: the profiled command's own exit code -- see the log tail)
  echo "pmc pass $i ($PMC) exit $rc; csv rows: $(cat $R/gpurun_out/pmc_${DT}_b${B}_c${LB}_p$i/*counter_collection.csv 2>/dev/null | wc -l)"
  tail -3 $R/gpurun_out/pmc_${DT}_b${B}_c${LB}_p$i.log | cut -c1-300
done
ls $R/gpurun_out/pmc_${DT}_b${B}_c${LB}_p1
