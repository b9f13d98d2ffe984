"""Multi-process (world_size=2, gloo, CPU) tests of the batch-shard host logic.  The forward
is injected (the float64 oracle stands in for the GPU kernels here: what is under test is the
partition / broadcast / gather plumbing of whenet_hip/shard.py, which is identical on RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch  # noqa: F401
import torch.multiprocessing as mp

from whenet_hip.shard import shard_bounds

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_shard_bounds_partition():
    for n in (0, 1, 2, 5, 7, 64, 65, 511, 512):
        for world in (1, 2, 3, 4, 8):
            got = [shard_bounds(n, world, r) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            for (a, b), (c, d) in zip(got, got[1:]):
                assert b == c
            sizes = [b - a for a, b in got]
            assert max(sizes) - min(sizes) <= 1
    assert [shard_bounds(512, 8, r) for r in range(8)] == [(64 * r, 64 * r + 64) for r in range(8)]
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    sys.path.insert(0, os.path.join(ROOT, "headposeestimation-whenet_amd"))
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from whenet_hip import synth, weights as W
    from whenet_hip.shard import ShardedWHENet, broadcast_bytes
    from oracle import whenet_oracle as O

    # snapshot broadcast: only rank 0 "has the file"
    blob = W.pack(W.synthetic(1234)) if rank == 0 else None
    blob = broadcast_bytes(blob, 0)
    w = W.unpack(blob)

    calls = []

    def fwd(u8):
        calls.append(u8.shape[0])
        r = O.forward(u8, w, np.float64)
        return np.stack([r["yaw"], r["pitch"], r["roll"]], 1).astype(np.float32), r["argmax"]

    m = ShardedWHENet(forward=fwd)
    crops = synth.scene_crops(n, seed=11)           # every rank builds the same global batch
    y, p, r = m.get_angle(crops)
    lo, hi = m.bounds(n)
    # the pre-sharded form (each rank hands over ITS shard, as bench.py / a per-GPU feeder does): same
    # rows; a rank whose shard is empty (n < world) must not call the forward at all
    before = len(calls)
    mine, am = m.forward_local(crops[lo:hi], global_batch=False)
    assert mine.shape == (hi - lo, 3) and am.shape == (hi - lo, 3)
    assert np.array_equal(mine[:, 0], y[lo:hi]) and np.array_equal(mine[:, 2], r[lo:hi])
    assert len(calls) == before + (1 if hi > lo else 0)
    del calls[before:]
    # float arrays holding byte values are converted (never reinterpreted); wrong shapes are refused
    yf, _, _ = m.get_angle(crops.astype(np.float32))
    assert np.array_equal(yf, y)
    del calls[before:]
    try:
        m.get_angle(np.zeros((n, 64, 64, 3), np.uint8))
        raise AssertionError("bad shape accepted")
    except ValueError:
        pass
    q.put((rank, W.checksum(w), calls, (lo, hi), np.stack([y, p, r], 1)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1, 3, 4])
def test_sharded_get_angle_matches_single_process(weights, n):
    from whenet_hip import synth, weights as W
    from oracle import whenet_oracle as O
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = O.forward(synth.scene_crops(n, seed=11), weights, np.float64)
    ref_ang = np.stack([ref["yaw"], ref["pitch"], ref["roll"]], 1).astype(np.float32)
    seen = 0
    for rank, csum, calls, (lo, hi), ang in sorted(res, key=lambda t: t[0]):
        assert csum == W.checksum(weights)                  # broadcast delivered the snapshot intact
        assert calls == ([hi - lo] if hi > lo else []) and (lo, hi) == shard_bounds(n, world, rank)
        seen += hi - lo
        assert ang.shape == (n, 3)
        assert np.array_equal(ang, ref_ang)                 # every rank holds the full, ordered result
    assert seen == n
