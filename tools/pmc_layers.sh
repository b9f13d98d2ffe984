RND=r06
mkdir -p gpurun_out/pmc profiles/$RND
for DT in f32s f16; do
  bash tools/pmc_round.sh $DT 64 64 > gpurun_out/pmc/pmc_${DT}_c64.log 2>&1; echo "pmc $DT exit $?"
  python tools/pmc_summary.py gpurun_out/pmc_${DT}_b64_c64_p gpurun_out/pmc/pmc_traffic_${DT}_b64.json 64 gpurun_out/pmc_${DT}_b64_c64_layers.json > gpurun_out/pmc/pmc_${DT}_b64_c64_by_kernel.txt 2>&1; echo "summary $DT exit $?"
  cp gpurun_out/pmc_${DT}_b64_c64_layers.json gpurun_out/pmc/
done
tail -60 gpurun_out/pmc/pmc_f32s_b64_c64_by_kernel.txt
