#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one directory per pass, prefix + 1..N) per kernel NAME: averages
over every dispatch of the run (the timed region and the profile pass launch the same sub-batch
geometry).  usage: pmc_summary.py gpurun_out/pmc_f16_b64_c64_p [traffic.json crops_per_launch]
The traffic JSON holds one set per crops-per-launch ("by_crops_per_launch"): an existing file is updated, not replaced."""
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

prefix = sys.argv[1]
CXXFILT = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
_cache = {}


def demangle_whenet(n):
    """rocprofv3 / llvm-cxxfilt leave the _Float16 instantiations mangled (DF16_): decode the
    simple template argument lists of this library's kernels by hand."""
    m = re.search(r"\d+(whenet_[a-z_0-9]+?)I((?:DF16_|f|L[ib]\d+E)+)E", n)
    if not m:
        return None
    args = []
    for tok in re.findall(r"DF16_|f|L[ib]\d+E", m.group(2)):
        if tok == "DF16_":
            args.append("_Float16")
        elif tok == "f":
            args.append("float")
        elif tok[1] == "b":
            args.append("true" if tok[2:-1] == "1" else "false")
        else:
            args.append(tok[2:-1])
    return f"{m.group(1)}<{', '.join(args)}>"


def short(n):
    if n not in _cache:
        d = demangle_whenet(n) if n.startswith("_Z") else None
        if d is None:
            m = re.search(r"(whenet_\w+(<[^(]*>)?)", n)
            d = m.group(1) if m else n[:60]
        _cache[n] = d
    return _cache[n]


agg = defaultdict(lambda: {"n": defaultdict(int), "c": defaultdict(float), "t": 0.0, "tn": 0})
for d in sorted(glob.glob(prefix + "[0-9]")):
    f = glob.glob(d + "/*counter_collection.csv")
    if not f:
        continue
    seen = set()
    for r in csv.DictReader(open(f[0])):
        if "whenet" not in r["Kernel_Name"]:
            continue
        k = short(r["Kernel_Name"])
        a = agg[k]
        a["c"][r["Counter_Name"]] += float(r["Counter_Value"])
        a["n"][r["Counter_Name"]] += 1
        key = (d, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            a["t"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            a["tn"] += 1


def avg(a, name):
    return a["c"][name] / a["n"][name] if a["n"][name] else 0.0


print(f"{'kernel':52s}{'us':>7s} {'act%':>6s} {'valu%':>6s} {'lds%':>5s} {'wait%':>6s} {'ldsconf%':>8s} {'fetchMB':>8s} {'writeMB':>8s} {'L2hit%':>6s} {'valu/wv':>8s} {'vmemRD/wv':>9s}")
out = {}
mfma_rows = []
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["t"]):
    wc = avg(a, "SQ_WAVE_CYCLES") or 1
    waves = avg(a, "SQ_WAVES") or 1
    hit, miss = avg(a, "TCC_HIT_sum"), avg(a, "TCC_MISS_sum")
    fetch = 2 * avg(a, "FETCH_SIZE") * 1024       # gfx950: FETCH_SIZE tallies 64 B per 128-B request on 16-B/lane streams
    write = avg(a, "WRITE_SIZE") * 1024
    print(f"{k:52s}{a['t'] / max(a['tn'], 1):7.1f} {100 * avg(a, 'SQ_ACTIVE_INST_ANY') / wc:6.1f} {100 * avg(a, 'SQ_ACTIVE_INST_VALU') / wc:6.1f} "
          f"{100 * avg(a, 'SQ_ACTIVE_INST_LDS') / wc:5.1f} {100 * avg(a, 'SQ_WAIT_ANY') / wc:6.1f} "
          f"{100 * avg(a, 'SQ_LDS_BANK_CONFLICT') / (avg(a, 'SQ_ACTIVE_INST_LDS') or 1):8.1f} {fetch / 1e6:8.2f} {write / 1e6:8.2f} "
          f"{100 * hit / ((hit + miss) or 1):6.1f} {avg(a, 'SQ_INSTS_VALU') / waves:8.0f} {avg(a, 'SQ_INSTS_VMEM_RD') / waves:9.1f}")
    busy_cu, mfma_busy = avg(a, "SQ_BUSY_CU_CYCLES"), avg(a, "SQ_VALU_MFMA_BUSY_CYCLES")
    mfma_rows.append((k, a["t"] / max(a["tn"], 1), mfma_busy, avg(a, "SQ_INSTS_VALU_MFMA_MOPS_F16"), avg(a, "SQ_INSTS_MFMA"),
                      busy_cu, avg(a, "GRBM_GUI_ACTIVE")))
    out[k] = {"mfma_busy_cycles": mfma_busy, "busy_cu_cycles": busy_cu,
              "mfma_util_of_busy_cu": (mfma_busy / (4 * busy_cu)) if busy_cu else None,
              "dispatches_seen": a["tn"], "avg_us_under_pmc": a["t"] / max(a["tn"], 1), "hbm_bytes_per_launch": fetch + write,
              "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
              "valu_active_pct_of_wave_cycles": 100 * avg(a, "SQ_ACTIVE_INST_VALU") / wc,
              "valu_insts_per_wave": avg(a, "SQ_INSTS_VALU") / waves,
              "wait_pct_of_wave_cycles": 100 * avg(a, "SQ_WAIT_ANY") / wc,
              "l2_hit_pct": 100 * hit / ((hit + miss) or 1),
              "mfma_busy_pct_of_cu_cycles": (100 * mfma_busy / (4 * busy_cu)) if busy_cu else None,
              "lds_bank_conflict_pct": 100 * avg(a, "SQ_LDS_BANK_CONFLICT") / (avg(a, "SQ_ACTIVE_INST_LDS") or 1)}
# Matrix-core occupancy per kernel (pass 6).  SQ_VALU_MFMA_BUSY_CYCLES counts cycles a SIMD's MFMA pipe is
# busy (32 per v_mfma_f32_32x32x16_f16), summed over the chip; SQ_BUSY_CU_CYCLES counts cycles a CU has a wave,
# summed over CUs; a CU has 4 SIMDs -> util = mfma_busy / (4 * busy_cu).
if any(r[2] for r in mfma_rows):
    print()
    print(f"{'kernel':52s}{'us':>7s} {'mfma_busy':>11s} {'mops_f16':>10s} {'insts_mfma':>10s} {'busy_cu':>11s} {'util/busyCU%':>12s}")
    for k, us, mb, mops, im, bc, gui in mfma_rows:
        print(f"{k:52s}{us:7.1f} {mb:11.0f} {mops:10.0f} {im:10.0f} {bc:11.0f} {100 * mb / (4 * bc) if bc else 0:12.2f}")
if len(sys.argv) > 2:
    cpl = str(int(sys.argv[3])) if len(sys.argv) > 3 else None
    if cpl is None:
        raise SystemExit("pmc_summary.py: the traffic JSON needs the crops per launch of the profiled chain (third argument)")
    blob = {}
    if os.path.exists(sys.argv[2]):
        try:
            blob = json.load(open(sys.argv[2]))
        except ValueError:
            blob = {}
    if "by_crops_per_launch" not in blob:
        blob = {"by_crops_per_launch": {}}
    blob["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KB -> bytes, FETCH_SIZE x2 "
                    "(MI355X_MICROARCH.md HBM section: 64 B tallied per 128-B request on 16-B/lane streams; consistent "
                    "here with the known read volumes of the expand and depthwise kernels); averages per dispatch; one "
                    "set per crops-per-launch of the profiled chain (ONE chain alone on the GPU: tools/pmc_round.sh)")
    blob["by_crops_per_launch"][cpl] = {"crops_per_launch": int(cpl), "source": prefix, "kernels": out}
    json.dump(blob, open(sys.argv[2], "w"), indent=1)
