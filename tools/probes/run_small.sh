ONLY=b13 timeout 200 ./tools/probes/front2_probe
WHENET_FRONT_THREADS=256 NOCHECK=1 timeout 300 ./tools/probes/front2_probe
