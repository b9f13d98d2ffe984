cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6c
V=${1:-v2}
timeout 300 python -m pytest tests/test_mb7.py -x -q -m gpu 2>&1 | tail -5
./tools/probes/mb7_probe | tee gpurun_out/r6c/mb7_probe_$V.txt
./tools/probes/mb7_probe_tile | grep -A1 "n= 64" | tee gpurun_out/r6c/mb7_probe_tile_$V.txt
