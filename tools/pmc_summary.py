#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one directory per pass) per kernel launch of ONE forward.
usage: pmc_summary.py gpurun_out/pmc_f16_b64_p   (prefix; passes _p1.._pN are merged)"""
import csv
import glob
import sys
from collections import defaultdict, OrderedDict

prefix = sys.argv[1]
per = OrderedDict()      # dispatch order key -> {counter: value}
for d in sorted(glob.glob(prefix + "[0-9]")):
    f = glob.glob(d + "/*counter_collection.csv")
    if not f:
        continue
    rows = [r for r in csv.DictReader(open(f[0])) if "whenet" in r["Kernel_Name"]]
    # group by dispatch id
    disp = OrderedDict()
    for r in rows:
        k = int(r["Dispatch_Id"])
        e = disp.setdefault(k, {"name": r["Kernel_Name"], "grid": r["Grid_Size"], "t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "c": {}})
        e["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    keys = sorted(disp)
    # last complete forward = last 66 whenet dispatches that start with a stem kernel
    stems = [i for i, k in enumerate(keys) if "stem" in disp[k]["name"] and i + 66 <= len(keys)]
    i0 = stems[-1]
    for j, k in enumerate(keys[i0:i0 + 66]):
        e = per.setdefault(j, {"name": disp[k]["name"], "grid": disp[k]["grid"], "c": {}, "t": disp[k]["t"]})
        e["c"].update(disp[k]["c"])


def short(n):
    n = n.split("whenet_")[1] if "whenet_" in n else n
    return n.split("(")[0][:44]


print(f"{'kernel':46s}{'us':>7s} {'busy%':>6s} {'valu%':>6s} {'lds%':>5s} {'wait%':>6s} {'ldsconf%':>8s} {'fetchMB':>8s} {'writeMB':>8s} {'L2hit%':>6s} {'valu/wv':>8s} {'vmemRD/wv':>9s}")
for j, e in per.items():
    c = defaultdict(float, e["c"])
    wc = c["SQ_WAVE_CYCLES"] or 1
    busy = c["SQ_BUSY_CYCLES"]
    waves = c["SQ_WAVES"] or 1
    hit, miss = c["TCC_HIT_sum"], c["TCC_MISS_sum"]
    print(f"{short(e['name']):46s}{e['t']:7.1f} {100 * c['SQ_ACTIVE_INST_ANY'] / wc:6.1f} {100 * c['SQ_ACTIVE_INST_VALU'] / wc:6.1f} "
          f"{100 * c['SQ_ACTIVE_INST_LDS'] / wc:5.1f} {100 * c['SQ_WAIT_ANY'] / wc:6.1f} {100 * c['SQ_LDS_BANK_CONFLICT'] / (c['SQ_ACTIVE_INST_LDS'] or 1):8.1f} "
          f"{2 * c['FETCH_SIZE'] / 1024:8.2f} {c['WRITE_SIZE'] / 1024:8.2f} {100 * hit / ((hit + miss) or 1):6.1f} {c['SQ_INSTS_VALU'] / waves:8.0f} {c['SQ_INSTS_VMEM_RD'] / waves:9.1f}")
