#!/usr/bin/env python3
"""Per-launch kernel durations of ONE forward from a rocprofv3 --kernel-trace CSV
(hardware timestamps, no event overhead): takes the last complete run of 66 whenet kernels."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "whenet" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last stem kernel that is followed by 65 more kernels
idx = [i for i, r in enumerate(rows) if "stem" in r["Kernel_Name"] and i + 66 <= len(rows)]
i0 = idx[-1 if len(sys.argv) < 3 else int(sys.argv[2])]
run = rows[i0:i0 + 66]
t0 = int(run[0]["Start_Timestamp"])
tot = 0
prev_end = t0
names = []
for r in run:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    short = n.split("whenet_")[1].split("(")[0][:46] if "whenet_" in n else n[:46]
    dur = (e - s) / 1e3
    gap = (s - prev_end) / 1e3
    tot += dur
    prev_end = e
    print(f"{short:48s} dur {dur:8.2f} us  gap {gap:6.2f} us  grid {r.get('Grid_Size_X','?'):>8s} wg {r.get('Workgroup_Size_X','?'):>5s} vgpr {r.get('VGPR_Count', r.get('Arch_VGPR_Count','?'))} lds {r.get('LDS_Block_Size','?')}")
print(f"sum kernel durations {tot:.1f} us; wall first-start..last-end {(int(run[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
