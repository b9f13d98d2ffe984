#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r05
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -x > $R/gpurun_out/r05/pytest_gpu_call1.txt 2>&1; echo "pytest exit $?"; tail -15 $R/gpurun_out/r05/pytest_gpu_call1.txt
timeout 400 python tools/fanout_sweep.py f16 > $R/gpurun_out/r05/fanout_sweep_f16.txt 2>&1; echo "fanout exit $?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $R/gpurun_out/r05/fanout_sweep_f16.txt | tail -80
bash tools/pmc_round.sh f16 512 256 > $R/gpurun_out/r05/pmc_c256.log 2>&1; echo "pmc 256 exit $?"; tail -5 $R/gpurun_out/r05/pmc_c256.log
python tools/pmc_summary.py gpurun_out/pmc_f16_b512_c256_p gpurun_out/r05/pmc_traffic_f16_b512.json 256 > $R/gpurun_out/r05/pmc_f16_b512_c256_by_kernel.txt 2>&1; echo "pmc summary exit $?"
head -70 $R/gpurun_out/r05/pmc_f16_b512_c256_by_kernel.txt
