#!/bin/bash
# PMC counter passes over a probe binary (separate passes; --pmc only with --kernel-trace).
#   bash tools/probe_pmc.sh <tag> <command...>      e.g.  ONLY=b5 NOCHECK=1 bash tools/probe_pmc.sh b5 ./tools/probes/front2_probe
# Prints, per kernel name and grid size, the per-dispatch averages of every counter.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
CMD="$@"
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/ppmc_${TAG}_p$i
  ( cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $R/gpurun_out/ppmc_${TAG}_p$i -o p -- $CMD > $R/gpurun_out/ppmc_${TAG}_p$i.log 2>&1 )
  echo "pmc pass $i exit $?; rows: $(cat $R/gpurun_out/ppmc_${TAG}_p$i/*counter_collection.csv 2>/dev/null | wc -l)"
done
python3 - "$R/gpurun_out/ppmc_${TAG}_p" <<'EOF'
import csv, glob, sys, re
from collections import defaultdict
prefix = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in sorted(glob.glob(prefix + "[0-9]")):
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            m = re.search(r"(whenet_\w+?)(I[^E]*E+)?v", n)
            short = re.sub(r"_ZN6whenet12_GLOBAL__N_1\d+", "", n)[:48]
            key = (short, r.get("Grid_Size", "?"), r.get("LDS_Block_Size", "?"))
            a = agg[key][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for key in sorted(agg):
    print(f"== {key[0]} grid {key[1]} lds {key[2]}")
    c = {k: v[0] / max(v[1], 1) for k, v in agg[key].items()}
    for k in sorted(c):
        print(f"   {k:32s} {c[k]:16.1f}")
    if "SQ_WAVES" in c and c["SQ_WAVES"] > 0:
        w = c["SQ_WAVES"]
        print(f"   per wave: cycles {4*c.get('SQ_WAVE_CYCLES',0)/w:.0f}  valu insts {c.get('SQ_INSTS_VALU',0)/w:.0f}  trans {c.get('SQ_INSTS_VALU_TRANS',0)/w:.0f}"
              f"  mfma {c.get('SQ_INSTS_MFMA',0)/w:.0f}  lds {c.get('SQ_INSTS_LDS',0)/w:.0f}  salu {c.get('SQ_INSTS_SALU',0)/w:.0f}  vmem rd {c.get('SQ_INSTS_VMEM_RD',0)/w:.0f} wr {c.get('SQ_INSTS_VMEM_WR',0)/w:.0f}")
        wc = c.get("SQ_WAVE_CYCLES", 0)
        if wc:
            print(f"   of wave cycles: active {100*c.get('SQ_ACTIVE_INST_ANY',0)/wc:.1f}%  valu {100*c.get('SQ_ACTIVE_INST_VALU',0)/wc:.1f}%  lds {100*c.get('SQ_ACTIVE_INST_LDS',0)/wc:.1f}%"
                  f"  wait_any {100*c.get('SQ_WAIT_ANY',0)/wc:.1f}%  wait_inst {100*c.get('SQ_WAIT_INST_ANY',0)/wc:.1f}%")
        if c.get("SQ_LDS_IDX_ACTIVE"):
            print(f"   lds: bank conflict cycles / idx active = {100*c.get('SQ_LDS_BANK_CONFLICT',0)/c['SQ_LDS_IDX_ACTIVE']:.1f}%")
        if c.get("SQ_BUSY_CU_CYCLES"):
            print(f"   mfma busy / (4 x busy CU cycles) = {100*c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(4*c['SQ_BUSY_CU_CYCLES']):.1f}%")
EOF
