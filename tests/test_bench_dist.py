"""bench.py's distributed bookkeeping with N > 1 (world_size 2, gloo, CPU): the SAME functions the GPU run uses
(bench.Comm / timed_steps / reduce_runs, whenet_hip.shard.shard_bounds) with an injected forward -- a sleep whose length
depends on the rank.  What is under test is what the driver reads from the line: barrier + max-over-ranks timing, the
same number of repeats on every rank, per-rank rates / crops / enqueue times gathered in rank order, and the weak /
strong partition of the batch.  No GPU, no scaling claim.  Also: the cgroup / NUMA host logic of the bench line."""
import importlib.util
import os
import socket
import sys
import time

import pytest
import torch  # noqa: F401
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _bench():
    spec = importlib.util.spec_from_file_location("whenet_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, strong, q):
    sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd"))
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bench = _bench()
    from whenet_hip.shard import shard_bounds
    comm = bench.Comm(True, torch.device("cpu"))
    assert (comm.world, comm.rank) == (world, rank)
    global_batch = 101                                   # ragged on purpose: 51 + 50
    if strong:
        lo, hi = shard_bounds(global_batch, world, rank)
        B, total = hi - lo, global_batch
    else:
        B, total = 64, world * 64
    steps, warmup = 5, 2
    calls = []
    per_step = 0.004 * (1 + rank)                         # rank 1 is the slow one: it sets the job's time

    def step(i):
        calls.append(i)
        time.sleep(per_step)

    synced = []
    runs = bench.timed_steps(step, lambda: synced.append(1), comm, steps, warmup)
    res = bench.reduce_runs(runs, comm, B, total, steps)
    q.put((rank, {"n_runs": len(runs), "calls": len(calls), "synced": len(synced), "el": res["el"], "els": res["els"],
                  "own": sorted(r[0] for r in runs), "value": res["value"], "per_rank_crops_s": res["per_rank_crops_s"],
                  "per_rank_crops": res["per_rank_crops"], "per_rank_enq": res["per_rank_enqueue_ms_per_step"],
                  "B": B, "total": total}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("strong", [False, True], ids=["weak", "strong"])
def test_timed_region_bookkeeping_world_size_2(strong):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, strong, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = got[0], got[1]
    steps, warmup = 5, 2
    # the K-step region is ~40 ms on the slow rank: repeated, and every rank repeats the same number of times
    assert r0["n_runs"] == r1["n_runs"] and r0["n_runs"] >= 3 and r0["n_runs"] % 2 == 1
    assert r0["calls"] == r1["calls"] == warmup + steps * r0["n_runs"]
    assert r0["synced"] == 1 + r0["n_runs"]
    # the job's time is the max over ranks, identical on both, and at least the slow rank's 5 x 8 ms
    assert r0["el"] == r1["el"] and r0["els"] == r1["els"]
    assert r0["el"] >= 5 * 0.008 * 0.95
    # ... while rank 0 alone would have been about twice as fast on its own clock (the barrier after the region is inside
    # the bracket, so its own elapsed time includes the wait: its ENQUEUE time shows its own pace)
    assert r0["per_rank_enq"][0] < 0.75 * r0["per_rank_enq"][1]
    assert r0["per_rank_enq"] == r1["per_rank_enq"] and len(r0["per_rank_enq"]) == 2
    # value = crops of ALL ranks per step x steps / that time
    total = 101 if strong else 128
    assert r0["total"] == total and abs(r0["value"] - total * steps / r0["el"]) < 1e-6 * r0["value"]
    assert r0["value"] == r1["value"]
    # per-rank gathers arrive in rank order on every rank
    assert r0["per_rank_crops"] == r1["per_rank_crops"] == ([51, 50] if strong else [64, 64])
    assert sum(r0["per_rank_crops"]) == total
    assert r0["per_rank_crops_s"] == r1["per_rank_crops_s"] and len(r0["per_rank_crops_s"]) == 2
    for r in (0, 1):
        own_median = got[r]["own"][len(got[r]["own"]) // 2]
        assert abs(r0["per_rank_crops_s"][r] - got[r]["B"] * steps / own_median) < 1e-6 * r0["per_rank_crops_s"][r]


def test_single_process_comm_is_a_no_op_group():
    bench = _bench()
    comm = bench.Comm(False, torch.device("cpu"))
    assert comm.world == 1 and comm.max_over_ranks(1.25) == 1.25 and comm.gather(3.0) == [3.0] and comm.gather_ints(7) == [7]
    n = []
    runs = bench.timed_steps(lambda i: n.append(i), lambda: None, comm, 4, 1, no_repeat=True)
    assert len(runs) == 1 and len(n) == 5
    r = bench.reduce_runs(runs, comm, 8, 8, 4)
    assert r["per_rank_crops"] == [8] and r["value"] > 0


def test_pmc_traffic_refuses_a_different_launch_geometry(tmp_path, monkeypatch):
    """roofline.traffic is only printed for a PMC set collected at the profiled launch's crops per launch."""
    import json
    bench = _bench()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    d = tmp_path / "profiles" / "r04"
    d.mkdir(parents=True)
    kern = "whenet_pw_splitk_kernel<_Float16, 2, 1, true, 0>"
    (d / "pmc_traffic_f16_b64.json").write_text(json.dumps(
        {"by_crops_per_launch": {"32": {"crops_per_launch": 32, "kernels": {kern: {"hbm_bytes_per_launch": 10.0e6}}}}}))
    t, src, _ = bench.pmc_traffic("f16", 64, 64, kern)
    assert t is None and "refused" in src
    t, src, _ = bench.pmc_traffic("f16", 64, 32, kern)
    assert t == 10.0e6 and "crops_per_launch 32" in src
    old = tmp_path / "profiles" / "r03"
    old.mkdir()
    (old / "pmc_traffic_f16_b64.json").write_text(json.dumps({"kernels": {kern: {"hbm_bytes_per_launch": 1.0}}}))
    t, src, _ = bench.pmc_traffic("f16", 64, 64, kern)
    assert t is None and "pre-round-4" in src


def test_gpu_numa_binding_host_logic(tmp_path):
    """whenet_hip.shard: NUMA node / CPU list of a GPU from a (fake) sysfs tree; two ranks sharing a node split its CPUs."""
    from whenet_hip import shard
    assert shard._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    bus = "0000:c1:00.0"
    dev = tmp_path / "bus" / "pci" / "devices" / bus
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    mine = sorted(os.sched_getaffinity(0))
    (node / "cpulist").write_text(",".join(str(c) for c in mine) + "\n")
    assert shard.gpu_numa_cpus(bus, str(tmp_path)) == (1, mine)
    (dev / "numa_node").write_text("-1\n")
    assert shard.gpu_numa_cpus(bus, str(tmp_path)) == (-1, [])
    assert shard.gpu_numa_cpus("0000:ff:00.0", str(tmp_path)) == (-1, [])
    info = shard.bind_rank_to_gpu_numa(bus, sysfs_root=str(tmp_path))
    assert info["bound"] is False and info["numa_node"] == -1 and sorted(os.sched_getaffinity(0)) == mine
    (dev / "numa_node").write_text("1\n")
    try:
        if len(mine) >= 2:
            info = shard.bind_rank_to_gpu_numa(bus, index_on_node=1, peers_on_node=2, sysfs_root=str(tmp_path))
            half = len(mine) // 2
            assert info["bound"] and info["cpus"] == half and sorted(os.sched_getaffinity(0)) == mine[half:2 * half]
            os.sched_setaffinity(0, mine)
        info = shard.bind_rank_to_gpu_numa(bus, sysfs_root=str(tmp_path))
        assert info["bound"] and info["cpus"] == len(mine) and info["numa_node"] == 1
    finally:
        os.sched_setaffinity(0, mine)


def test_cgroup_quota_and_measured_parallelism():
    bench = _bench()
    q = bench.cgroup_cpu_quota()
    assert q is None or q > 0
    par = bench.measured_parallelism(2, 0.3)
    # structural only: a timing ratio on a shared box is not a test oracle (by_work read 3.2 for 2 processes)
    assert par is not None and par["processes"] == 2
    assert 0.0 < par["by_work"] <= 2.0 and 0.0 < par["by_cpu_seconds"] <= 2.0
