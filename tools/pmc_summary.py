#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one directory per pass, prefix + 1..N) per kernel NAME: averages
over every dispatch of the run (the timed region and the profile pass launch the same sub-batch
geometry).  usage: pmc_summary.py gpurun_out/pmc_f16_b64_c64_p [traffic.json crops_per_launch [layers.json]]
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


def new_agg():
    return defaultdict(lambda: {"n": defaultdict(int), "c": defaultdict(float), "t": 0.0, "tn": 0, "kernel": ""})


agg = new_agg()
# round 6: the same counters per LAYER.  With the chain's launch list (bench.py --dump-layers of the SAME command: fifth argument)
# every forward of a pass is the same sequence of launches, so a dispatch's position in its forward names its layer -- kernels that
# several layers share (front.hip's float instantiation runs b2 and b6; the split-K GEMM five blocks) get a row per layer.
layer_names, layer_kernels = [], []
if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
    for l in json.load(open(sys.argv[4]))["launches"]:
        if l.get("kind") != "calib":
            layer_names.append(l["layer"])
            layer_kernels.append(l["kernel"])
lagg = new_agg()
layer_note = None
for d in sorted(glob.glob(prefix + "[0-9]")):
    f = glob.glob(d + "/*counter_collection.csv")
    if not f:
        continue
    seen = set()
    rows_by_dispatch = defaultdict(list)
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
        if "whenet_empty_kernel" not in r["Kernel_Name"]:
            rows_by_dispatch[int(r["Dispatch_Id"])].append(r)
    if layer_names:
        order = sorted(rows_by_dispatch)
        L = len(layer_names)
        base = lambda n: n.replace(" ", "").split("<")[0].split("EPK")[0]
        if len(order) % L != 0:
            layer_note = f"{d}: {len(order)} chain dispatches are not a multiple of the {L} launches of a forward -- no per-layer rows"
            continue
        for i, did in enumerate(order):
            rs = rows_by_dispatch[did]
            li = i % L
            if base(short(rs[0]["Kernel_Name"])) != base(layer_kernels[li]):
                layer_note = f"{d}: dispatch {did} is {short(rs[0]['Kernel_Name'])}, the launch list says {layer_kernels[li]} -- no per-layer rows"
                lagg = new_agg()
                layer_names = []
                break
            a = lagg[layer_names[li]]
            a["kernel"] = short(rs[0]["Kernel_Name"])
            for r in rs:
                a["c"][r["Counter_Name"]] += float(r["Counter_Value"])
                a["n"][r["Counter_Name"]] += 1
            a["t"] += (int(rs[0]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])) / 1e3
            a["tn"] += 1


def avg(a, name):
    return a["c"][name] / a["n"][name] if a["n"][name] else 0.0


print(f"{'kernel':52s}{'us':>7s} {'act%':>6s} {'valu%':>6s} {'lds%':>5s} {'wait%':>6s} {'ldsconf%':>8s} {'fetchMB':>8s} {'writeMB':>8s} {'L2hit%':>6s} {'valu/wv':>8s} {'vmemRD/wv':>9s}")
def row_of(a):
    wc = avg(a, "SQ_WAVE_CYCLES") or 1
    waves = avg(a, "SQ_WAVES") or 1
    hit, miss = avg(a, "TCC_HIT_sum"), avg(a, "TCC_MISS_sum")
    fetch = 2 * avg(a, "FETCH_SIZE") * 1024
    write = avg(a, "WRITE_SIZE") * 1024
    busy_cu, mfma_busy = avg(a, "SQ_BUSY_CU_CYCLES"), avg(a, "SQ_VALU_MFMA_BUSY_CYCLES")
    return {"kernel": a.get("kernel", ""), "dispatches_seen": a["tn"], "avg_us_under_pmc": a["t"] / max(a["tn"], 1),
            "hbm_bytes_per_launch": fetch + write, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
            "valu_active_pct_of_wave_cycles": 100 * avg(a, "SQ_ACTIVE_INST_VALU") / wc,
            "valu_insts_per_wave": avg(a, "SQ_INSTS_VALU") / waves,
            "wait_pct_of_wave_cycles": 100 * avg(a, "SQ_WAIT_ANY") / wc,
            "l2_hit_pct": 100 * hit / ((hit + miss) or 1),
            "mfma_busy_pct_of_cu_cycles": (100 * mfma_busy / (4 * busy_cu)) if busy_cu else None,
            "lds_bank_conflict_pct": 100 * avg(a, "SQ_LDS_BANK_CONFLICT") / (avg(a, "SQ_ACTIVE_INST_LDS") or 1)}


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
layers_out = {}
if lagg:
    alg = {}
    if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
        alg = {l["layer"]: l for l in json.load(open(sys.argv[4]))["launches"]}
    print()
    print(f"{'layer':14s}{'us':>7s} {'valu%':>6s} {'wait%':>6s} {'fetchMB':>8s} {'writeMB':>8s} {'algMB':>7s} {'x alg':>6s} {'L2hit%':>6s} {'mfma%':>6s} {'valu/wv':>8s}  kernel")
    for name in layer_names:
        r = row_of(lagg[name])
        ab = alg.get(name, {}).get("alg_bytes", 0.0)
        r["alg_bytes_per_launch"] = ab
        r["traffic_over_alg_bytes"] = (r["hbm_bytes_per_launch"] / ab) if ab else None
        layers_out[name] = r
        print(f"{name:14s}{r['avg_us_under_pmc']:7.1f} {r['valu_active_pct_of_wave_cycles']:6.1f} {r['wait_pct_of_wave_cycles']:6.1f} "
              f"{r['fetch_bytes_per_launch'] / 1e6:8.2f} {r['write_bytes_per_launch'] / 1e6:8.2f} {ab / 1e6:7.2f} "
              f"{(r['traffic_over_alg_bytes'] or 0):6.2f} {r['l2_hit_pct']:6.1f} {(r['mfma_busy_pct_of_cu_cycles'] or 0):6.2f} "
              f"{r['valu_insts_per_wave']:8.0f}  {r['kernel'][:70]}")
if layer_note:
    print("\nper-layer rows: " + layer_note)
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
    if layers_out:
        blob["by_crops_per_launch"][cpl]["layers"] = layers_out
    json.dump(blob, open(sys.argv[2], "w"), indent=1)
