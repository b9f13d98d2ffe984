#!/usr/bin/env python3
"""Benchmark of the WHENet HIP path on MI355X -- prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype f16|f32]

A "step" is one pass of the hot path (uint8 crops resident in HBM -> yaw/pitch/roll,
argmax, logits) over one batch of B synthetic crops per GPU.  Default workload:
BASELINE.json configs[2] -- batch=64, 224x224, fp16 on one MI355X (the configuration the
"crops/sec at batch 64" metric is quoted on); at N>1 the same per-GPU batch on every rank
(configs[3], weak scaling, no data-path collective).  For N>1 the driver launches this file
under torch.distributed.run (one rank per GPU, RCCL); run directly with --gpus N>1 it
re-launches itself that way.

Schedule of the timed region: K forwards of the batch, `--inflight` (default 3) of them in flight per
GPU.  A forward is a chain of 46-51 dependent kernel launches (about half of a batch-64 forward is
launch latency), so a serving process keeps a few independent batches in flight; the library does
that with engine option "inflight" (n engines behind one handle, round-robin, every forward in
flight writes its own output buffers).  All K forwards complete inside the timed region (handle sync
+ torch.cuda.synchronize() + barrier on both sides).  `serial_schedule` carries the same K steps
run strictly one after the other (--inflight 1) for reference.

Short timed regions: when the K timed steps take less than 200 ms (the driver runs --steps 20 --warmup 5: ~10 ms),
the K-step region is repeated (each repeat again bracketed by sync + barrier) until ~0.3 s have been timed;
`ms_per_step` / `value` are the MEDIAN repeat, every repeat is listed in `repeats_ms_per_step` (`steps` stays K).

Extra objects on the line:
  sweep         the other north-star batch sizes in f16 -- 1, 8, 512 crops per step -- and the f32 parity configuration at batch
                64 (`f32_b64`), ~0.1 s each: `value` (3 forwards in flight), `value_serial`, and the dominant kernel's roofline
                fraction at that batch (for the crops per launch whenet_profile() reports)
  roofline      dominant kernel: algorithmic bytes / HIP-event duration, per launch; `traffic` = PMC FETCH_SIZE + WRITE_SIZE per launch
                from the committed set collected at the SAME crops per launch (profiles/rNN/pmc_traffic_*.json; null otherwise)
                (whenet_profile(): one event between consecutive launches on the chain's
                stream, ONE forward of the batch alone on the GPU, eager pass run right after
                the timed region -- the figures rocprofv3's kernel trace also reports, since
                tracing serialises the overlapped forwards) vs 8 TB/s HBM
  cpu_baseline  the float32 torch-CPU restatement of the reference path ("port": the true
                Keras path cannot run here), timed on this box's host cores (P pinned processes x T threads
                covering half of the logical CPUs; `cores` = what the run could use: min(threads launched, cgroup CPU quota,
                measured parallelism), `threads_launched` = P*T), rank 0, N=1
  latency_b1    configs[1]: batch=1 fp32 single-crop latency (median / p99), N=1 only
  frame_pipeline  configs[4]: one video frame + k head boxes per submission (PCIe included), N=1 only
  pcie_inclusive  the host-pointer forms on the same batch, H2D + D2H included (never `value`)
  yolo_postprocess  §8f row 4: whenet_yolo_eval on one 416x416 detector output, per call, host to host
  serial_schedule the timed region with one forward at a time (also as top-level `value_serial` /
                `ms_per_step_serial`)
  config.per_rank_crops_s / world_size / backend   what each rank did on its own clock, the RCCL world

--strong: a fixed global batch (--global-batch, default 512) split over the ranks ("scaling": "strong");
the default is weak scaling (the same --batch on every rank).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch  # first: one HIP runtime in the process (its libamdhip64 is shared with libwhenet_hip)

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "headposeestimation-whenet_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
PATH_BOUND_CROPS_S = {          # BASELINE.md §2, per GPU, f16, 6.29 TB/s
    "layer_granular": 227280.0, "mbconv_2kernel_fusion": 453209.0}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400,
                    help="timed steps (default 400: >= 200 ms timed at batch 64)")
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--batch", type=int, default=64, help="crops per GPU per step")
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32", "f32s"])
    ap.add_argument("--profile-iters", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 fp32 latency leg")
    ap.add_argument("--repeat", type=int, default=1, help="debug: issue every kernel launch this many times")
    ap.add_argument("--inflight", type=int, default=3,
                    help="independent forwards in flight per GPU (engine option 'inflight': 1 = strictly one "
                         "after the other; n > 1 = n engines used round-robin, each forward as one chain)")
    ap.add_argument("--no-serial", action="store_true", help="skip the serial-schedule reference region")
    ap.add_argument("--lanes", type=int, default=0, help="concurrent sub-batch chains per forward (0 = engine default)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="engine option passed to whenet_set_option (e.g. trunk=0)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: a fixed GLOBAL batch (--global-batch, default 512) split over the ranks")
    ap.add_argument("--global-batch", type=int, default=512, help="global batch of --strong")
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
    ap.add_argument("--dump-layers", default="", help="write the per-launch profile (JSON) to this path")
    ap.add_argument("--no-sweep", action="store_true", help="skip the batch 1 / 8 / 512 sweep")
    ap.add_argument("--no-repeat", action="store_true", help="time the K steps once even if that is < 200 ms")
    ap.add_argument("--check-crops", type=int, default=16, help="crops of the batch compared with the float64 oracle")
    return ap.parse_args()


def relaunch(args) -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def _cpu_worker(idx, cpus, threads, secs, barrier, q):
    """One pinned process of the CPU baseline: the reference-faithful path on its own CPU set."""
    try:
        os.sched_setaffinity(0, cpus)
    except (AttributeError, OSError):
        pass
    os.environ["OMP_NUM_THREADS"] = str(threads)
    import torch as T
    T.set_num_threads(threads)
    from whenet_hip import weights as W, synth
    from oracle.whenet_torch import TorchWHENet
    m = TorchWHENet(W.synthetic(1234))
    crops = synth.noise_crops(8, seed=idx)
    m.get_angle(crops.copy(), batch_size=8)                  # warm-up
    barrier.wait(timeout=300)
    done, t0 = 0, time.perf_counter()
    while True:
        m.get_angle(crops.copy(), batch_size=8)
        done += 8
        el = time.perf_counter() - t0
        if el >= secs:
            break
    q.put((idx, done, el))


def cgroup_cpu_quota():
    """CPU quota of this process's cgroup in cores (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us), walking up
    from the process's own group; None = unlimited or unreadable."""
    paths = []
    try:
        with open("/proc/self/cgroup") as f:
            for line in f:
                _, ctrl, rel = line.strip().split(":", 2)
                rel = rel.lstrip("/")
                if ctrl == "":                                      # v2 unified hierarchy
                    d = os.path.join("/sys/fs/cgroup", rel)
                    while True:
                        paths.append(("v2", os.path.join(d, "cpu.max")))
                        if os.path.normpath(d) == "/sys/fs/cgroup":
                            break
                        d = os.path.dirname(d)
                elif "cpu" in ctrl.split(","):                      # v1 cpu controller
                    d = os.path.join("/sys/fs/cgroup", ctrl, rel)
                    paths.append(("v1", d))
                    paths.append(("v1", os.path.join("/sys/fs/cgroup", ctrl)))
    except (OSError, ValueError):
        pass
    paths += [("v2", "/sys/fs/cgroup/cpu.max"), ("v1", "/sys/fs/cgroup/cpu"), ("v1", "/sys/fs/cgroup/cpu,cpuacct")]
    best = None
    for kind, pth in paths:
        try:
            if kind == "v2":
                with open(pth) as f:
                    q, per = f.read().split()[:2]
                if q == "max":
                    continue
                cores = float(q) / float(per)
            else:
                with open(os.path.join(pth, "cpu.cfs_quota_us")) as f:
                    q = int(f.read())
                with open(os.path.join(pth, "cpu.cfs_period_us")) as f:
                    per = int(f.read())
                if q <= 0:
                    continue
                cores = q / per
        except (OSError, ValueError, ZeroDivisionError):
            continue
        best = cores if best is None else min(best, cores)
    return best


_SPIN = ("import sys,time\n"
         "t0=float(sys.argv[1]); d=float(sys.argv[2])\n"
         "while time.time()<t0: pass\n"
         "n=0; c=time.process_time()\n"
         "while time.time()<t0+d:\n"
         "    for _ in range(20000): n+=1\n"
         "print(n, time.process_time()-c)\n")


def measured_parallelism(nproc: int, dur: float = 1.0):
    """How many CPUs this container really gets: `nproc` interpreter processes spin over the same wall-clock window; the
    sum of their loop counts over one process's count alone (and the sum of their CPU seconds over the window) is the
    effective core count whatever sched_getaffinity / nproc claim (CPU quotas and steal do not show up there)."""
    def run(k):
        t0 = time.time() + 0.5 + 0.012 * k
        ps = [subprocess.Popen([sys.executable, "-c", _SPIN, repr(t0), repr(dur)], stdout=subprocess.PIPE, text=True)
              for _ in range(k)]
        n, cpu = 0, 0.0
        for pr in ps:
            o = pr.communicate(timeout=120)[0].split()
            n += int(o[0])
            cpu += float(o[1])
        return n, cpu
    try:
        # the single-process rate is the yardstick: best of three windows (interpreter warm-up and clock ramp only ever
        # make one run SLOWER, which would inflate the ratio), and the ratio cannot exceed the process count
        n1 = max(run(1)[0] for _ in range(3))
        nk, cpuk = run(nproc)
        return {"processes": nproc, "by_work": min(nk / max(n1, 1), float(nproc)), "by_cpu_seconds": min(cpuk / dur, float(nproc))}
    except (OSError, ValueError, IndexError, subprocess.SubprocessError):
        return None


def cpu_baseline(seconds: float):
    """Reference-faithful CPU path (oracle/whenet_torch.py: float64 normalise, batch_size=8 chunks as whenet.py:27,
    numpy decode) on the host cores of this box.  One torch process does not scale over a many-core host (batch-8
    convolutions: 8 threads are its best), so the baseline runs P processes x T = 8 threads, each pinned to its own
    CPUs (sched_setaffinity), covering HALF of the logical CPUs (the other half are mostly SMT siblings), all timed
    over the same >= 5 s window: `value` = crops of all processes / the longest window.
    Round 4: `cores` is what the run could actually USE -- min(threads launched, cgroup CPU quota, measured
    parallelism of this container) -- next to `threads_launched`; round 3 printed 128 where 16 processes gave 1.15 x
    one process (`scaling_anomaly`)."""
    import multiprocessing as mp
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        avail = list(range(os.cpu_count() or 1))
    ncpu = len(avail)
    T = min(8, ncpu)
    P = max(1, (ncpu // 2) // T)
    secs = max(5.0, seconds * 0.3)
    ctx = mp.get_context("spawn")

    def run(nproc):
        barrier = ctx.Barrier(nproc)
        q = ctx.Queue()
        procs = []
        for i in range(nproc):
            cpus = avail[i * T:(i + 1) * T]
            pr = ctx.Process(target=_cpu_worker, args=(i, cpus, T, secs, barrier, q))
            pr.start()
            procs.append(pr)
        res = [q.get(timeout=600) for _ in range(nproc)]
        for pr in procs:
            pr.join(timeout=60)
        return sum(r[1] for r in res), max(r[2] for r in res)

    d1, e1 = run(1)
    dP, eP = (d1, e1) if P == 1 else run(P)
    quota = cgroup_cpu_quota()
    par = measured_parallelism(min(P * T, 128))
    launched = int(P * T)
    eff = float(launched)
    if quota is not None:
        eff = min(eff, quota)
    if par is not None:
        eff = min(eff, max(par["by_work"], par["by_cpu_seconds"]))
    rate1, rateP = d1 / e1, dP / eP
    best, best_threads = (rateP, launched) if rateP >= rate1 else (rate1, T)
    anomaly = bool(P > 1 and rateP < 0.5 * P * rate1)
    return {"value": best, "unit": "crops/s", "cores": int(round(min(eff, best_threads))), "kind": "port",
            "threads_launched": launched, "effective_cores": round(eff, 1),
            "cgroup_cpu_quota_cores": quota, "measured_parallelism": par,
            "scaling_anomaly": anomaly,
            "scaling_note": (f"{P} processes gave {rateP / rate1:.2f} x one process ({rateP:.1f} vs {rate1:.1f} crops/s): the host "
                             f"does not deliver {launched} cores to this container; `cores` is the effective figure") if anomaly else None,
            "sample": f"{dP} crops as batches of 8 (whenet.py:27 batch_size=8) in {eP:.1f} s by {P} pinned processes x "
                      f"{T} threads; torch-CPU f32 restatement of whenet.py:22-34 incl. float64 normalise + numpy decode",
            "multi_process": {"value": rateP, "processes": P, "threads_each": T},
            "single_process": {"value": rate1, "unit": "crops/s", "cores": int(T),
                               "sample": f"{d1} crops in {e1:.1f} s, one process, {T} threads"},
            "host_cpus": os.cpu_count() or ncpu, "usable_cpus": ncpu}


def frame_leg(h):
    """demo_video.py:49-58 as FramePipeline runs it: 720p synthetic frame, k in {1,4,16} heads.
    `sync_*` = submit + collect of one frame (per-frame latency); `pipelined_frames_per_s` = depth-2
    overlap of frame i+1's staging with frame i's GPU work."""
    from whenet_hip import frames as F, synth

    class _M:                      # FramePipeline only needs the handle
        _handle = h

    frame = synth.video_frame()
    res = {"frame": "1280x720 BGR uint8 (2.76 MB over PCIe per frame)", "dtype": "f16"}
    for k in (1, 4, 16):
        boxes = synth.head_boxes(k)
        fp = F.FramePipeline(_M, depth=2)
        lat = []
        for i in range(260):
            a = time.perf_counter()
            fp.process(frame, boxes)
            lat.append(time.perf_counter() - a)
        lat = np.array(lat[60:]) * 1e6
        a = time.perf_counter()
        n = 0
        for i in range(200):
            if fp.in_flight == 2:
                fp.collect()
                n += 1
            fp.submit(frame, boxes)
        while fp.in_flight:
            fp.collect()
            n += 1
        el = time.perf_counter() - a
        res[f"k{k}"] = {"sync_median_us": float(np.median(lat)), "sync_p99_us": float(np.percentile(lat, 99)),
                        "pipelined_frames_per_s": n / el, "pipelined_heads_per_s": n * k / el}
    return res


def host_leg(h, crops):
    """Host-pointer forms, H2D of the crops (150,528 B each) and D2H of the results included:
    blocking whenet_forward_u8 (pageable numpy memory), and whenet_submit_u8 / whenet_collect with
    the handle's engines (pinned staging, forwards of successive batches overlapping)."""
    B = crops.shape[0]
    for _ in range(5):
        h.forward(crops)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 0.5:
        h.forward(crops)
        n += 1
    blocking = n * B / (time.perf_counter() - t0)
    pend = []
    t0 = time.perf_counter()
    done = 0
    for i in range(120):
        if len(pend) == 3:
            h.collect(pend.pop(0), B)
            done += 1
        pend.append(h.submit(crops))
    while pend:
        h.collect(pend.pop(0), B)
        done += 1
    piped = done * B / (time.perf_counter() - t0)
    return {"batch": B, "dtype": "f16", "forward_u8_blocking_crops_s": blocking,
            "submit_collect_3_in_flight_crops_s": piped,
            "note": "host uint8 crops in, angles out; H2D 9.6 MB per batch of 64 over PCIe included"}


def dropin_leg(blob, device):
    """The reference's own call shapes through the drop-in CLASS (python, ctypes, PCIe, everything included):
    get_angle(uint8[1,224,224,3]) per head (demo.py:14, demo_video.py:27) as wall-clock latency, f32 (the parity
    configuration) and f16; get_angle(uint8[512,...]) as host->host throughput (whenet.py:22-27 takes any N; large
    batches are cut into chunks over the handle's engines, include/whenet_hip.h "fanout_min")."""
    import whenet
    rng = np.random.default_rng(0)
    big = rng.integers(0, 256, (512, 224, 224, 3), dtype=np.uint8)
    out = {}
    for dd in ("f32", "f32s", "f16"):        # f32s is the class default (whenet.WHENet(snapshot) with no dtype)
        with whenet.WHENet(snapshot=blob, dtype=dd, device=device) as m:
            one = big[:1]
            lat = []
            for _ in range(700):
                a = time.perf_counter()
                m.get_angle(one)
                lat.append(time.perf_counter() - a)
            lat = np.array(lat[100:]) * 1e6
            out[f"get_angle_b1_{dd}"] = {"median_us": float(np.median(lat)), "p99_us": float(np.percentile(lat, 99)), "iters": 600}
            for _ in range(3):
                m.get_angle(big)
            t0 = time.perf_counter()
            k = 0
            while time.perf_counter() - t0 < (0.8 if dd == "f16" else 0.5):
                m.get_angle(big)
                k += 1
            out[f"get_angle_b512_{dd}_crops_s"] = k * 512 / (time.perf_counter() - t0)
    out["default_dtype"] = "f32s"
    out["note"] = "whenet.WHENet(...).get_angle(numpy uint8): python + ctypes + H2D + forward + D2H, pageable host memory; the class default is f32s"
    return out


def yolo_leg(h):
    """yolo_eval (yolo_v3/model.py:193-232) for one 416x416 detector output (10,647 boxes, one
    class), host maps in -> detections out (H2D of 255 KB included), next to the numpy restatement on the host."""
    from whenet_hip import synth
    from oracle import yolo_oracle as Y
    maps = synth.yolo_maps(1, num_classes=1)
    kw = dict(max_boxes=20, score_threshold=0.3, iou_threshold=0.45)
    for _ in range(5):
        got = h.yolo_eval(maps, synth.YOLO_ANCHORS, 1, (720, 1280), **kw)
    lat = []
    for _ in range(200):
        a = time.perf_counter()
        h.yolo_eval(maps, synth.YOLO_ANCHORS, 1, (720, 1280), **kw)
        lat.append(time.perf_counter() - a)
    a = time.perf_counter()
    for _ in range(3):
        ref = Y.yolo_eval(maps, synth.YOLO_ANCHORS, 1, (720, 1280), **kw)
    cpu = (time.perf_counter() - a) / 3
    return {"boxes": 10647, "detections": int(len(got[0])), "same_count_as_numpy": bool(len(got[0]) == len(ref[0])),
            "gpu_median_us": float(np.median(lat) * 1e6), "numpy_port_us": cpu * 1e6,
            "note": "blocking whenet_yolo_eval per frame: H2D of the three maps, decode + NMS kernels, D2H"}


SWISH_CYCLES_PER_WAVE_VALUE = 24.0      # mul, exp, add, rcp, mul: 3 x 2.6 + 2 x 8.1 cycles per wave-instruction (docs/experiments.md 3.4)
SIMDS, SHADER_GHZ = 1024, 2.25


def swish_counts():
    """Swish evaluations per crop of every launch that carries them, by layer name (algorithmic: no halo recompute)."""
    from whenet_hip import spec
    out = {"stem": 112 * 112 * 32, "head": 49 * 1280, "stem+b1/dw": 2 * 112 * 112 * 32}
    for b in spec.blocks():
        if b.has_expand:
            out[f"b{b.index}/front"] = (b.h_in * b.h_in + b.h_out * b.h_out) * b.cexp
            out[f"b{b.index}/expand"] = b.h_in * b.h_in * b.cexp
        out[f"b{b.index}/dw"] = b.h_out * b.h_out * b.cexp
    return out


def summarise_profile(stats, crops_per_launch=None):
    """per-kernel totals of one forward's launch list + the dominant kernel's roofline object (HBM roof, and the
    VALU roof for kernels whose floor is the Swish activations: 2 quarter-rate transcendentals per value).
    crops_per_launch: what whenet_profile() says each launch processed (ABI 3: `crops` of every entry -- with option
    inflight > 1 a forward is ONE chain of the whole batch; round 3 assumed two lanes here and halved the VALU floor)."""
    if crops_per_launch is None:
        crops_per_launch = stats[0]["crops"]
    boundary_us = 0.0
    if stats and stats[-1]["kind"] == "calib":
        boundary_us = stats[-1]["avg_us"]
        stats = stats[:-1]
    sw = swish_counts()
    by_kernel = {}
    for st in stats:
        k = by_kernel.setdefault(st["kernel"], {"us": 0.0, "bytes": 0.0, "flops": 0.0, "launches": 0, "kind": st["kind"],
                                                "swish": 0.0})
        k["us"] += st["avg_us"]
        k["bytes"] += st["alg_bytes"]
        k["flops"] += st["alg_flops"]
        k["launches"] += 1
        k["swish"] += sw.get(st["layer"], 0) * crops_per_launch
    # the dominant kernel is the dominant LAUNCH (one layer of the chain), not the largest sum over a template name: round 5's
    # line flipped between block 2's front kernel (one 64 us launch) and the 7 x 7 project GEMM (five 13 us launches)
    top = max(stats, key=lambda st: st["avg_us"])
    dom_name = top["kernel"]
    dom = {"us": top["avg_us"], "bytes": top["alg_bytes"], "flops": top["alg_flops"], "launches": 1, "kind": top["kind"],
           "swish": sw.get(top["layer"], 0) * crops_per_launch, "layer": top["layer"]}
    achieved = dom["bytes"] / (dom["us"] * 1e-6) / 1e9
    valu_floor_us = dom["swish"] / 64.0 * SWISH_CYCLES_PER_WAVE_VALUE / SIMDS / (SHADER_GHZ * 1e3)
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "kernel": dom_name, "layer": dom["layer"], "launches_per_step": dom["launches"], "avg_launch_us": dom["us"] / dom["launches"],
            "selection": "the launch of the chain with the largest average duration (by layer)",
            "alg_bytes_per_launch": dom["bytes"] / dom["launches"],
            "tflops": dom["flops"] / (dom["us"] * 1e-6) / 1e12,
            "valu": {"what": "the kernel's Swish activations alone on the 1024 SIMDs: 5 VALU instructions per value, 2 of "
                             "them quarter-rate (v_exp_f32, v_rcp_f32) = 24 cycles per wave-instruction group at 2.25 GHz; "
                             "this, not HBM, is the binding roof of the fused expand+depthwise kernels (DESIGN.md 6.2)",
                     "swish_values_per_launch": dom["swish"] / dom["launches"],
                     "floor_us_per_launch": valu_floor_us / dom["launches"],
                     "frac": (valu_floor_us / dom["us"]) if (dom["us"] > 0 and dom["swish"] > 0) else None,
                     "fused_front_kernels": None}}
    # the same VALU roof for the class of kernels it binds: all fused expand+depthwise launches of the forward
    fr = [k for k in by_kernel.values() if k["kind"] == "front"]
    if fr:
        fus, fsw = sum(k["us"] for k in fr), sum(k["swish"] for k in fr)
        ffloor = fsw / 64.0 * SWISH_CYCLES_PER_WAVE_VALUE / SIMDS / (SHADER_GHZ * 1e3)
        roof["valu"]["fused_front_kernels"] = {"launches": sum(k["launches"] for k in fr), "us": fus,
                                                "swish_floor_us": ffloor, "frac": ffloor / fus if fus > 0 else None}
    return stats, by_kernel, dom_name, dom, roof, boundary_us


def sweep_leg(blob, local_rank, dev, lanes_opt):
    """The other north-star batch sizes (BASELINE.json: batch 1 / 8 / 64 / 512, f16) and the f32 parity configuration at
    batch 64: ~0.1 s timed per schedule."""
    from whenet_hip import _lib, synth
    res = {}

    def key_of(nb, dt):
        return f"b{nb}" if dt == _lib.F16 else (f"f32s_b{nb}" if dt == _lib.F32S else f"f32_b{nb}")

    for nb, dt in ((1, _lib.F16), (8, _lib.F16), (512, _lib.F16), (64, _lib.F32), (64, _lib.F32S)):
        h = _lib.Handle(blob, device=local_rank, dtype=dt)
        if lanes_opt > 0:
            h.set_option("lanes", lanes_opt)
        crops = synth.noise_crops(nb, seed=100 + nb)
        d_crops = torch.from_numpy(crops).to(dev)
        bufs = [(torch.zeros((nb, 3), dtype=torch.float32, device=dev), torch.zeros((nb, 3), dtype=torch.int32, device=dev),
                 torch.zeros((nb, 252), dtype=torch.float32, device=dev)) for _ in range(3)]

        def region(nslots, steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                y, a, l = bufs[i % nslots]
                h.forward_device(d_crops.data_ptr(), nb, y.data_ptr(), a.data_ptr(), l.data_ptr())
            h.sync()
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        entry = {}
        for nslots, key in ((1, "serial"), (3, "inflight3")):
            if nslots == 3:
                h.set_option("inflight", 3)
                if lanes_opt > 0:
                    h.set_option("lanes", lanes_opt)
            region(nslots, 6)                                        # warm-up (graph capture)
            probe = region(nslots, 6) / 6
            steps = int(min(2000, max(6, 0.1 / max(probe, 1e-5))))
            els = sorted(region(nslots, steps) for _ in range(3))
            entry[key] = {"crops_s": nb * steps / els[1], "ms_per_step": els[1] / steps * 1e3, "steps": steps}
        stats = h.profile(d_crops.data_ptr(), nb, 3 if dt == _lib.F16 else 10)
        cpl = stats[0]["crops"]
        _, _, dname, dtop, roof, _ = summarise_profile(stats)
        res[key_of(nb, dt)] = {"value": entry["inflight3"]["crops_s"], "value_serial": entry["serial"]["crops_s"],
                         "ms_per_step": entry["inflight3"]["ms_per_step"], "ms_per_step_serial": entry["serial"]["ms_per_step"],
                         "steps": entry["inflight3"]["steps"], "dtype": {_lib.F16: "f16", _lib.F32: "f32", _lib.F32S: "f32s"}[dt],
                         "dominant_kernel": {"kernel": roof["kernel"], "avg_launch_us": roof["avg_launch_us"],
                                             "crops_per_launch": stats[0]["crops"],
                                             "frac_hbm": roof["frac"], "frac_valu": roof["valu"]["frac"]}}
        if dt != _lib.F16:
            # the parity-grade configurations carry a roofline object of their own (same method as the headline's)
            dts = "f32" if dt == _lib.F32 else "f32s"
            tr, tr_src, tr_extra = pmc_traffic(dts, nb, cpl, dname, dtop["layer"])
            roof.update({"traffic": tr, "traffic_source": tr_src, "pmc": tr_extra, "crops_per_launch": cpl,
                         "traffic_over_alg_bytes": (tr / roof["alg_bytes_per_launch"]) if tr else None,
                         "chain_us_per_step": sum(st["avg_us"] for st in stats if st["kind"] != "calib")})
            res[key_of(nb, dt)]["roofline"] = roof
        h.close()
    # f32s: parity of THIS run's f32s forward against the float64 oracle on the driver's 16 check crops (the 512-crop contract is
    # tests/test_f32s.py)
    res["note"] = ("crops resident in HBM; b1 / b8 / b512 = f16, f32_b64 = the parity-grade configuration (1e-3 deg, exact "
                   "argmax) at the headline batch, f32s_b64 = the same float32 storage with the 1x1 products as binary16 hi/lo pairs "
                   "on the f16 matrix cores (WHENET_F32S: same 1e-3 deg bar, tests/test_f32s.py); value = 3 forwards in flight, value_serial = one at a time; ~0.1 s "
                   "timed per schedule (median of 3); dominant kernel from whenet_profile() at that batch (one chain of "
                   "`crops_per_launch` crops)")
    return res


class Comm:
    """What the timed region needs from the process group: a fence (barrier + device sync), the max of a host float
    over the ranks, an all-gather of one host float per rank.  `distributed` False = one process, no group.
    The device is whatever the group's backend moves (cuda for "nccl" = RCCL, cpu for "gloo" in tests/test_bench_dist.py,
    which runs exactly these functions with world_size 2 and an injected forward)."""

    def __init__(self, distributed: bool, device, device_sync=None):
        self.distributed = distributed
        self.device = device
        self.device_sync = device_sync or (lambda: None)
        if distributed:
            import torch.distributed as dist
            self.dist = dist
            self.world, self.rank = dist.get_world_size(), dist.get_rank()
        else:
            self.dist, self.world, self.rank = None, 1, 0

    def fence(self):
        if self.distributed:
            self.dist.barrier()
        self.device_sync()

    def max_over_ranks(self, x: float) -> float:
        t = torch.tensor([x], dtype=torch.float64, device=self.device)
        if self.distributed:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, x: float):
        if not self.distributed:
            return [float(x)]
        g = [torch.zeros(1, dtype=torch.float64, device=self.device) for _ in range(self.world)]
        self.dist.all_gather(g, torch.tensor([x], dtype=torch.float64, device=self.device))
        return [float(v.item()) for v in g]

    def gather_ints(self, x: int):
        return [int(round(v)) for v in self.gather(float(x))]


def timed_steps(step, sync, comm: Comm, steps: int, warmup: int, no_repeat: bool = False):
    """W untimed + exactly K timed steps, bracketed by comm.fence() (barrier + device synchronize) on both sides; when
    the K steps take < 200 ms the same bracketed region is repeated (every rank the same number of times: the decision
    is taken on the max over ranks) and every repeat is returned: [(elapsed, host_enqueue_time)] of THIS rank."""
    def region():
        comm.fence()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        enq = time.perf_counter() - t0          # host time to enqueue the K steps (must stay below the elapsed time)
        sync()
        comm.fence()
        return time.perf_counter() - t0, enq

    for i in range(warmup):
        step(i)
    sync()
    runs = [region()]
    el0 = comm.max_over_ranks(runs[0][0])       # every rank takes the same number of repeats
    if el0 < 0.2 and not no_repeat:
        extra = min(40, max(2, int(0.3 / max(el0, 1e-4))))
        extra += extra % 2                      # odd number of regions in total: the median is a measured one
        for _ in range(extra):
            runs.append(region())
    return runs


def reduce_runs(runs, comm: Comm, crops_this_rank: int, total_per_step: int, steps: int):
    """max over ranks per repeat -> the median repeat is the job's time; per-rank figures on each rank's own clock."""
    els = sorted(comm.max_over_ranks(r[0]) for r in runs)
    el = els[len(els) // 2]
    own = sorted(r[0] for r in runs)
    el_rank = own[len(own) // 2]
    enq = sorted(r[1] for r in runs)[len(runs) // 2]
    return {"el": el, "els": els, "value": total_per_step * steps / el,
            "per_rank_crops_s": comm.gather(crops_this_rank * steps / el_rank),
            "per_rank_enqueue_ms_per_step": comm.gather(enq / steps * 1e3),
            "per_rank_crops": comm.gather_ints(crops_this_rank),
            "enqueue_ms_per_step": enq / steps * 1e3}


def numa_bind(comm: Comm, local_rank: int):
    """One rank per GPU: pin the process to the CPUs of its GPU's NUMA node (whenet_hip/shard.py), ranks that share a
    node take disjoint slices of it.  Single-process runs are left alone (the driver's N=1 line must not depend on
    it); WHENET_BIND_NUMA=0 / 1 forces it off / on."""
    from whenet_hip.shard import bind_rank_to_gpu_numa, gpu_numa_cpus, gpu_pci_bus_id
    want = os.environ.get("WHENET_BIND_NUMA")
    if want == "0" or (want != "1" and comm.world == 1):
        return {"bound": False, "reason": "single process" if want != "0" else "WHENET_BIND_NUMA=0"}
    try:
        bus = gpu_pci_bus_id(local_rank)
    except (RuntimeError, AssertionError, AttributeError) as e:
        return {"bound": False, "reason": f"no PCI id: {e}"[:100]}
    node, _ = gpu_numa_cpus(bus)
    # peers = ranks on the same NUMA node OF THE SAME HOST (a node number means nothing across hosts)
    import zlib
    host = zlib.crc32(os.uname().nodename.encode()) & 0x3fffff
    keys = comm.gather_ints(host * 64 + (node + 1))
    same = [r for r, k in enumerate(keys) if k == keys[comm.rank]]
    return bind_rank_to_gpu_numa(bus, index_on_node=same.index(comm.rank), peers_on_node=len(same))


def pmc_traffic(dtype: str, B: int, crops_per_launch: int, dom_name: str, dom_layer: str = ""):
    """HBM traffic of the dominant kernel from the committed PMC passes (tools/pmc_round.sh): only a set collected at
    the SAME crops per launch as the profile ran is comparable with `alg_bytes_per_launch`; otherwise traffic is null
    (round 3 printed a 32-crop figure next to 64-crop algorithmic bytes)."""
    tried = []
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        rel = os.path.join("profiles", rnd, f"pmc_traffic_{dtype}_b{B}.json")
        try:
            with open(os.path.join(ROOT, rel)) as f:
                blob = json.load(f)
        except (OSError, ValueError):
            continue
        sets = blob.get("by_crops_per_launch")
        if sets is None:
            tried.append(f"{rel}: no crops_per_launch recorded (pre-round-4 file) -- refused")
            continue
        tk = sets.get(str(crops_per_launch))
        if tk is None:
            tried.append(f"{rel}: holds crops_per_launch {sorted(sets)} -- not {crops_per_launch}, refused")
            continue
        cands = []
        lay = (tk.get("layers") or {}).get(dom_layer)
        if lay is not None:                       # round 6: counters per LAYER (tools/pmc_summary.py with the chain's launch list)
            cands.append((dom_name, lay))
        cands += list(tk["kernels"].items())
        for name, v in cands:
            if name.replace(" ", "") == dom_name.replace(" ", ""):
                extra = {k2: v[k2] for k2 in ("valu_active_pct_of_wave_cycles", "valu_insts_per_wave", "mfma_busy_pct_of_cu_cycles",
                                              "waves_per_simd", "lds_bank_conflict_pct", "wait_pct_of_wave_cycles",
                                              "fetch_bytes_per_launch", "write_bytes_per_launch") if k2 in v} or None
                src = (f"{rel} [crops_per_launch {crops_per_launch}]: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) / WRITE_SIZE in "
                       f"separate passes (tools/pmc_round.sh) of ONE chain of {crops_per_launch} crops run alone (the counters "
                       f"are device-wide: with chains side by side a kernel's figures include the others' traffic); the "
                       f"same launch geometry whenet_profile() timed; committed file -- NOT re-measured by this run" +
                       (f"; counters of layer {dom_layer} alone" if v is lay else "; averaged over every launch of this kernel name"))
                return v["hbm_bytes_per_launch"], src, extra
        tried.append(f"{rel}: kernel {dom_name} not in the set")
    return None, ("no PMC set matches the profiled launch (" + "; ".join(tried) + ")") if tried else None, None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        sys.exit(relaunch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("WHENET_FORCE_DIST") == "1"   # the latter: 1-rank test of the RCCL path
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "RANK" not in os.environ:               # WHENET_FORCE_DIST=1 without a launcher: a 1-rank group
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            sk.close()
            os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"
        dist.init_process_group("nccl", device_id=dev)
    comm = Comm(distributed, dev, torch.cuda.synchronize)
    affinity = numa_bind(comm, local_rank)

    from whenet_hip import _lib, synth, weights as W
    from whenet_hip.shard import broadcast_bytes

    # weights: rank 0 builds the seeded synthetic snapshot; RCCL broadcast to the others
    blob = W.pack(W.synthetic(1234)) if rank == 0 else None
    if distributed:
        blob = broadcast_bytes(blob, 0, dev)
    dt = {"f16": _lib.F16, "f32": _lib.F32, "f32s": _lib.F32S}[args.dtype]
    h = _lib.Handle(blob, device=local_rank, dtype=dt)
    if args.no_graph:
        h.set_option("graph", 0)
    if args.repeat > 1:
        h.set_option("repeat", args.repeat)
    if args.lanes > 0:
        h.set_option("lanes", args.lanes)
    for kv in args.opt:
        k, v = kv.split("=")
        h.set_option(k, int(v))

    B = args.batch
    if args.strong:
        # fixed global batch split over the ranks (whenet_hip/shard.py's partition): per-GPU work shrinks with N
        from whenet_hip.shard import shard_bounds
        lo, hi = shard_bounds(args.global_batch, world, rank)
        B = hi - lo
    M = max(1, args.inflight)
    crops = synth.noise_crops(B, seed=rank)        # default_rng(seed) uint8
    d_crops = torch.from_numpy(crops).to(dev)
    # every forward in flight writes its own output buffers
    outs = [(torch.zeros((B, 3), dtype=torch.float32, device=dev), torch.zeros((B, 3), dtype=torch.int32, device=dev),
             torch.zeros((B, 252), dtype=torch.float32, device=dev)) for _ in range(M)]
    d_ypr, d_am, d_lg = outs[0]

    def timed(nslots):
        def step(i):
            y, a, l = outs[i % nslots]
            h.forward_device(d_crops.data_ptr(), B, y.data_ptr(), a.data_ptr(), l.data_ptr())
        return timed_steps(step, h.sync, comm, args.steps, args.warmup, args.no_repeat)

    total_per_step = args.global_batch if args.strong else world * B      # crops all ranks process per step

    serial = None
    if M > 1:
        if not args.no_serial:
            # the strictly serial schedule first (one forward at a time, 2 sub-batch lanes), for reference
            r1 = reduce_runs(timed(1), comm, B, total_per_step, args.steps)
            serial = {"value": r1["value"], "ms_per_step": r1["el"] / args.steps * 1e3, "in_flight": 1,
                      "repeats": len(r1["els"])}
        h.set_option("inflight", M)
        if args.lanes > 0:
            h.set_option("lanes", args.lanes)
    res = reduce_runs(timed(M), comm, B, total_per_step, args.steps)
    el, els, value = res["el"], res["els"], res["value"]

    # ---- per-kernel roofline (HIP events around every launch, same stream, eager) ----------
    stats = h.profile(d_crops.data_ptr(), B, args.profile_iters)
    if args.dump_layers and rank == 0:
        with open(args.dump_layers, "w") as f:
            json.dump({"batch": B, "dtype": args.dtype, "launches": stats}, f, indent=1)
    # A launch's figure is the event-to-event time on the chain's stream.  Back-to-back launches pipeline, so for
    # a real kernel that IS its duration as rocprofv3's hardware timestamps report it (profiles/: the
    # kernel-trace averages agree within a few %); only for an EMPTY kernel is the ~2-9 us event/dispatch gap
    # exposed.  The chain's last entry is such an empty kernel: reported as `boundary_us`, never subtracted.
    crops_per_launch, chains = stats[0]["crops"], stats[0]["chains"]     # what whenet_profile() actually ran
    stats, by_kernel, dom_name, dom, roofline, boundary_us = summarise_profile(stats)
    for st in stats:
        st["raw_us"] = st["avg_us"]
    # HBM traffic of that kernel from the PMC counters (FETCH_SIZE / WRITE_SIZE, collected in separate
    # rocprofv3 --pmc passes of this same command and committed under profiles/): per launch, bytes
    traffic, traffic_source, pmc_extra = pmc_traffic(args.dtype, B, crops_per_launch, dom_name, dom["layer"])
    roofline.update({"traffic": traffic, "traffic_source": traffic_source, "pmc": pmc_extra,
                "traffic_over_alg_bytes": (traffic / roofline["alg_bytes_per_launch"]) if traffic else None,
                "crops_per_launch": crops_per_launch, "chains_profiled": chains,
                "method": "one hipEvent between consecutive launches on the chain's stream; one forward of the batch "
                          "alone on the GPU (eager pass right after the timed region), as rocprofv3's kernel trace "
                          "sees it; the timed region overlaps `forwards_in_flight` such chains",
                "empty_kernel_event_to_event_us": boundary_us,
                "chain_us_per_step": sum(st["raw_us"] for st in stats),
                "by_kernel": {k: {"us": round(v["us"], 2), "launches": v["launches"],
                                  "GBps": round(v["bytes"] / (v["us"] * 1e-6) / 1e9, 1),
                                  "TFLOPs": round(v["flops"] / (v["us"] * 1e-6) / 1e12, 2),
                                  "valu_frac": (round(v["swish"] / 64.0 * SWISH_CYCLES_PER_WAVE_VALUE / SIMDS /
                                                      (SHADER_GHZ * 1e3) / v["us"], 3) if v["swish"] > 0 else None)}
                              for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]["us"])}})

    out = {
        "metric": "head crops/sec (224x224)", "value": value, "unit": "crops/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
        "repeats": len(els), "repeats_ms_per_step": [round(x / args.steps * 1e3, 4) for x in els],
        "timed_region_note": (f"{len(els)} repeats of the {args.steps}-step region (each bracketed by sync + barrier); value "
                              "and ms_per_step are the median repeat") if len(els) > 1 else "one region of K steps",
        "host_enqueue_ms_per_step": res["enqueue_ms_per_step"],
        "value_serial": serial["value"] if serial else (value if M == 1 else None),
        "ms_per_step_serial": serial["ms_per_step"] if serial else (el / args.steps * 1e3 if M == 1 else None),
        "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": (f"global batch {total_per_step} split over {world} GPU(s) (strong scaling), " if args.strong
                                else f"batch={B}/GPU ") +
                               f"224x224 uint8 crops resident in HBM -> angles+argmax+logits "
                               f"(BASELINE.json configs[{2 if world == 1 else 3}])",
                   "batch_per_gpu": B, "global_batch": total_per_step, "weights": "synthetic random-init seed 1234",
                   "parallelism": f"batch-shard x{world}, no data-path collective",
                   "world_size": world, "backend": ("nccl (RCCL)" if distributed else "none (single process)"),
                   "per_rank_crops_s": [round(x, 1) for x in res["per_rank_crops_s"]],
                   "per_rank_crops": res["per_rank_crops"],
                   "per_rank_host_enqueue_ms_per_step": [round(x, 4) for x in res["per_rank_enqueue_ms_per_step"]],
                   "cpu_affinity": affinity,
                   "graph": not args.no_graph,
                   "engine_options": dict(o.split("=", 1) for o in args.opt) or "defaults",
                   "launches_per_forward": h.info().n_kernels_per_forward,
                   "forwards_in_flight": M,
                   "schedule": (f"{M} independent forwards of the batch in flight per GPU (engine option inflight={M}: "
                                f"{M} engines round-robin, own streams/arena/graphs, one chain each); ms_per_step = "
                                "timed region / K, not the latency of one forward") if M > 1 else
                               "one forward at a time, up to 2 concurrent sub-batch chains inside it"},
        "roofline": roofline,
        "serial_schedule": serial,
        "path_fraction": {"per_gpu_crops_s": value / world,
                          "vs_layer_granular_bound": value / world / PATH_BOUND_CROPS_S["layer_granular"],
                          "vs_2kernel_fusion_bound": value / world / PATH_BOUND_CROPS_S["mbconv_2kernel_fusion"],
                          "bounds_crops_s": PATH_BOUND_CROPS_S},
    }

    if rank == 0 and world == 1:
        # correctness spot check against the float64 oracle (outside every timed region)
        from oracle import whenet_oracle as O
        nchk = max(1, min(args.check_crops, B))
        idx = np.unique(np.linspace(0, B - 1, nchk).astype(int))      # spread over the batch (every lane)
        ref = O.forward(crops[idx], W.synthetic(1234), np.float64)
        ref_ang = np.stack([ref["yaw"], ref["pitch"], ref["roll"]], axis=1)
        got = d_ypr.cpu().numpy()[idx]
        if any(not (torch.equal(y, d_ypr) and torch.equal(a, d_am) and torch.equal(l, d_lg)) for y, a, l in outs[1:min(M, args.steps)]):
            # every engine (that ran) must have produced the same bits: say which slot is off, against a fresh forward
            kept = [l.clone() for _, _, l in outs]
            h.sync()
            h.forward_device(d_crops.data_ptr(), B, d_ypr.data_ptr(), d_am.data_ptr(), d_lg.data_ptr())
            h.sync()
            torch.cuda.synchronize()
            msg = []
            for si, l in enumerate(kept[:min(M, args.steps)]):
                bad = torch.nonzero((l != d_lg).any(dim=1)).flatten().tolist()
                if bad:
                    msg.append(f"slot {si}: {len(bad)} crops {bad[:12]} max |logit diff| {float((l - d_lg).abs().max()):.4g}")
            raise AssertionError("in-flight forwards differ; against a fresh forward: " + ("; ".join(msg) or "all slots equal it"))
        err = np.abs(got - ref_ang)
        flips = int((d_am.cpu().numpy()[idx] != ref["argmax"]).sum())
        out["check"] = {"max_abs_deg_vs_f64_oracle": float(err.max()), "p95_abs_deg": float(np.percentile(err, 95)),
                        "mean_abs_deg": float(err.mean()), "crops": int(len(idx)),
                        "argmax_flips": flips, "bins": int(3 * len(idx)),
                        # north_star: angles within 1e-3 deg of the reference AND bin argmax exactly equal
                        "meets_north_star_parity": bool(err.max() <= 1e-3 and flips == 0),
                        "north_star_parity_note": ("binary16 activations cannot meet 1e-3 deg / exact argmax (weight rounding alone "
                                                   "moves angles by ~0.3 deg); the f32 configuration does -- sweep.f32_b64 carries its "
                                                   "throughput, latency_b1 its latency") if args.dtype == "f16" else None}
        if world == 1 and args.dtype == "f16" and not args.no_sweep:
            # the two parity-grade configurations on the SAME check crops (host call, outside every timed region)
            par = {}
            for name, pdt in (("f32", _lib.F32), ("f32s", _lib.F32S)):
                with _lib.Handle(blob, device=local_rank, dtype=pdt) as hp:
                    y, am, _ = hp.forward(np.ascontiguousarray(crops[idx]))
                e = np.abs(y - ref_ang)
                fl = int((am != ref["argmax"]).sum())
                par[name] = {"max_abs_deg_vs_f64_oracle": float(e.max()), "mean_abs_deg": float(e.mean()), "argmax_flips": fl,
                             "meets_north_star_parity": bool(e.max() <= 1e-3 and fl == 0)}
            out["check"]["parity_configurations"] = par
    if rank == 0 and world == 1 and not args.no_latency:
        # configs[1]: batch=1 fp32 latency
        h1 = _lib.Handle(blob, device=local_rank, dtype=_lib.F32)
        lat = []
        for i in range(1100):
            torch.cuda.synchronize()
            a = time.perf_counter()
            h1.forward_device(d_crops.data_ptr(), 1, d_ypr.data_ptr(), d_am.data_ptr(), d_lg.data_ptr())
            h1.sync()
            lat.append(time.perf_counter() - a)
        lat = np.array(lat[100:]) * 1e6
        out["latency_b1"] = {"dtype": "f32", "median_us": float(np.median(lat)), "p99_us": float(np.percentile(lat, 99)),
                             "iters": 1000, "crops_per_s": float(1e6 / np.median(lat))}
        h1.close()
        # the same single-crop call on the other parity-grade configuration (float32 storage, split products: WHENET_F32S)
        h1 = _lib.Handle(blob, device=local_rank, dtype=_lib.F32S)
        lat = []
        for i in range(600):
            torch.cuda.synchronize()
            a = time.perf_counter()
            h1.forward_device(d_crops.data_ptr(), 1, d_ypr.data_ptr(), d_am.data_ptr(), d_lg.data_ptr())
            h1.sync()
            lat.append(time.perf_counter() - a)
        lat = np.array(lat[100:]) * 1e6
        out["latency_b1"]["f32s"] = {"median_us": float(np.median(lat)), "p99_us": float(np.percentile(lat, 99)), "iters": 500}
        h1.close()
        # configs[4]: one video frame = one submission (host BGR frame + k YOLO boxes -> pinned copy,
        # H2D, crop/resize on the device, forward of the k heads, D2H), PCIe-inclusive, f16
        out["frame_pipeline"] = frame_leg(h)
        # PCIe-inclusive rates of the host-pointer forms on the same batch (never `value`)
        out["pcie_inclusive"] = host_leg(h, crops)
        # the drop-in class on the reference's call shapes (B=1 wall latency; one 512-crop call), never `value`
        out["dropin"] = dropin_leg(blob, local_rank)
        out["yolo_postprocess"] = yolo_leg(h)
    if rank == 0 and world == 1 and not distributed and not args.no_sweep and args.dtype == "f16" and B == 64 and not args.strong:
        out["sweep"] = sweep_leg(blob, local_rank, dev, args.lanes)
    if "sweep" in out:
        # the parity-grade figure as a first-class number: the faster of the two float32-storage configurations (both hold the
        # north_star bar: <= 1e-3 deg, exact argmax -- tests/test_gpu_parity.py, tests/test_f32s.py) at the headline batch
        pk = max(("f32_b64", "f32s_b64"), key=lambda k: out["sweep"][k]["value"])
        pe = out["sweep"][pk]
        out["parity_value"], out["parity_dtype"] = pe["value"], pe["dtype"]
        out["parity_value_serial"] = pe["value_serial"]
        out["parity_roofline"] = pe.get("roofline")
        out["parity_note"] = (f"sweep.{pk}: batch 64, crops resident in HBM, 3 forwards in flight (parity_value_serial: one at a time); "
                              "`value` above is the binary16 configuration BASELINE.json configs[2] names, which does NOT meet the 1e-3 deg bar")
        out["config"]["workload"] += (f"; value = {M} forwards of the batch in flight ({value:.0f} crops/s), value_serial = one forward at a "
                                      f"time ({out['value_serial']:.0f} crops/s); parity-grade ({pe['dtype']}): {pe['value']:.0f} crops/s")
    elif out.get("value_serial") is not None:
        out["config"]["workload"] += (f"; value = {M} forwards of the batch in flight ({value:.0f} crops/s), value_serial = one forward at a "
                                      f"time ({out['value_serial']:.0f} crops/s)")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    h.close()
    if rank == 0:
        try:                      # RCCL's version banner sits in the C stdio buffer: push it out first, so that the
            import ctypes         # JSON line is the LAST line of stdout
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
