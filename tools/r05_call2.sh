#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r05
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -x > $R/gpurun_out/r05/pytest_gpu_call2.txt 2>&1; echo "pytest exit $?"; tail -8 $R/gpurun_out/r05/pytest_gpu_call2.txt
timeout 400 python tools/host_path_probe.py > $R/gpurun_out/r05/host_path_probe.txt 2>&1; echo "probe exit $?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" $R/gpurun_out/r05/host_path_probe.txt | tail -60
