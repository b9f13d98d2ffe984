for s in b2 b3 b4; do ONLY=$s NOCHECK=1 bash tools/probe_pmc.sh $s ./tools/probes/front2_probe > gpurun_out/ppmc_$s.txt 2>&1; done
