"""Single-process multi-GPU front end: WHENet(snapshot, devices=[...]).
CPU: the split / ordering / threading / error logic over injected handles.  GPU: devices=[0, 0] (two handles on the one
GPU the test box has) must be bitwise one handle, ragged N included.  No scaling is claimed from a one-GPU box."""
import threading

import numpy as np
import pytest

from whenet_hip import _lib, synth
from whenet_hip.multi import MultiDeviceHandle
from whenet_hip.shard import shard_bounds


class FakeHandle:
    """Deterministic stand-in: 'angles' are per-crop byte statistics, so order and content are both checked."""
    made = []

    def __init__(self, snapshot, device=0, dtype=0):
        self.device, self.dtype = device, dtype
        self.threads, self.calls, self.options, self.closed = set(), [], [], False
        self.threads.add(threading.get_ident())
        self.fail_on = None
        FakeHandle.made.append(self)

    def forward(self, crops, want_logits=True):
        self.threads.add(threading.get_ident())
        self.calls.append(crops.shape[0])
        if self.fail_on is not None and crops.shape[0] == self.fail_on:
            raise ValueError("boom")
        f = crops.reshape(crops.shape[0], -1).astype(np.float64)
        ypr = np.stack([f.mean(1), f[:, ::7].mean(1), f.max(1)], axis=1).astype(np.float32)
        am = np.stack([f.argmax(1) % 120, f.argmin(1) % 66, (f.sum(1) % 66)], axis=1).astype(np.int32)
        lg = np.tile(ypr, (1, 84)).astype(np.float32) if want_logits else None
        return ypr, am, lg

    forward_f32 = forward

    def set_option(self, k, v):
        self.threads.add(threading.get_ident())
        self.options.append((k, v))

    def info(self):
        return "info"

    def yolo_eval(self, *a, **k):
        self.threads.add(threading.get_ident())
        return ("yolo", self.device)

    def close(self):
        self.threads.add(threading.get_ident())
        self.closed = True


@pytest.fixture
def fake():
    FakeHandle.made = []
    return FakeHandle


@pytest.mark.parametrize("ndev", [1, 2, 3, 8])
@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 64, 129])
def test_split_order_and_content(fake, ndev, n):
    crops = np.random.default_rng(n).integers(0, 256, (n, 224, 224, 3), dtype=np.uint8)
    m = MultiDeviceHandle(b"snap", list(range(ndev)), dtype=1, handle_factory=fake)
    want = FakeHandle(b"", 0).forward(crops)
    got = m.forward(crops, True)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    y, a, lg = m.forward(crops, False)
    assert lg is None and np.array_equal(y, want[0])
    # the shards are shard_bounds(): contiguous, sizes differ by at most one, empty shards are not launched
    world = min(ndev, n)
    sizes = [shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0] for r in range(world)]
    for h, s in zip(fake.made[:ndev], sizes + [None] * ndev):
        assert h.calls == ([s, s] if s else [])
    m.close()
    assert all(h.closed for h in fake.made[:ndev])


def test_one_thread_per_handle_and_options_reach_every_device(fake):
    m = MultiDeviceHandle(b"snap", [0, 1, 2], dtype=0, handle_factory=fake)
    crops = np.zeros((30, 224, 224, 3), np.uint8)
    for _ in range(3):
        m.forward(crops)
    m.set_option("inflight", 3)
    assert m.yolo_eval(1, 2) == ("yolo", 0) and m.info() == "info"
    m.close()
    hs = fake.made[:3]
    assert all(len(h.threads) == 1 for h in hs), "a handle was touched by more than one thread"
    assert len({next(iter(h.threads)) for h in hs}) == 3 and threading.get_ident() not in {next(iter(h.threads)) for h in hs}
    assert all(h.options == [("inflight", 3)] for h in hs)


def test_errors_propagate_after_every_shard_finished(fake):
    m = MultiDeviceHandle(b"snap", [0, 1], dtype=0, handle_factory=fake)
    fake.made[1].fail_on = 5
    with pytest.raises(ValueError, match="boom"):
        m.forward(np.zeros((10, 224, 224, 3), np.uint8))
    assert fake.made[0].calls == [5] and fake.made[1].calls == [5]
    fake.made[1].fail_on = None
    assert m.forward(np.zeros((10, 224, 224, 3), np.uint8))[0].shape == (10, 3)       # still usable
    m.close()
    m.close()                                                                              # idempotent
    with pytest.raises(ValueError):
        MultiDeviceHandle(b"snap", [], dtype=0, handle_factory=fake)


def test_a_failed_construction_closes_every_handle_on_its_own_thread(fake):
    """Round-5 advice: when one device's handle cannot be created, the handles the OTHER futures created -- including those that
    finished after the failure was seen -- are closed on the thread that made them, and the error is the first one."""
    def factory(snapshot, device=0, dtype=0):
        if device == 1:
            raise OSError("no such device")
        return FakeHandle(snapshot, device=device, dtype=dtype)

    with pytest.raises(OSError, match="no such device"):
        MultiDeviceHandle(b"snap", [0, 1, 2, 3], dtype=0, handle_factory=factory)
    assert sorted(h.device for h in fake.made) == [0, 2, 3]
    assert all(h.closed and len(h.threads) == 1 for h in fake.made)


def test_min_shard_keeps_small_batches_on_one_device(fake):
    m = MultiDeviceHandle(b"snap", [0, 1, 2, 3], dtype=0, handle_factory=fake, min_shard=16)
    m.forward(np.zeros((20, 224, 224, 3), np.uint8))
    assert [h.calls for h in fake.made[:4]] == [[20], [], [], []]
    m.forward(np.zeros((40, 224, 224, 3), np.uint8))
    assert [h.calls for h in fake.made[:4]] == [[20, 20], [20], [], []]
    m.close()


# ----------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_gpu_two_handles_on_one_device_are_bitwise_one_handle(dtype):
    import whenet
    crops = np.concatenate([synth.scene_crops(200, seed=5), synth.noise_crops(77, seed=6)])       # 277: ragged, >= fan-out
    with whenet.WHENet(dtype=dtype) as one, whenet.WHENet(dtype=dtype, devices=[0, 0]) as two:
        for n in (1, 2, 7, 64, 277):
            a = one.get_angle(crops[:n])
            la, aa = one.last_logits.copy(), one.last_argmax.copy()
            b = two.get_angle(crops[:n])
            for x, y in zip(a, b):
                assert x.dtype == np.float32 and np.array_equal(x, y), (dtype, n)
            assert np.array_equal(la, two.last_logits) and np.array_equal(aa, two.last_argmax)
        x = crops[:5].astype(np.float64) * 0.5 + 3.25                                            # real-valued input: forward_f32
        for u, v in zip(one.get_angle(x), two.get_angle(x)):
            assert np.array_equal(u, v)
        two.model.summary()


@pytest.mark.gpu
def test_gpu_large_batch_fanout_is_bitwise_one_forward():
    """get_angle(np.uint8[N >= 256]) is cut into 128-crop forwards over the handle's engines (capi.cpp, option fanout_min):
    same bits as the single forward, ragged tail, every staging mode (2 = round 6's default: the caller's array registered for the
    call, one host thread, copies ordered across the engines) / depth / chunk size; errors leave the handle usable."""
    crops = np.concatenate([synth.scene_crops(200, seed=15), synth.noise_crops(77, seed=16)])      # 277
    from whenet_hip import weights as W
    blob = W.pack(W.synthetic(1234))
    with _lib.Handle(blob, device=0, dtype=_lib.F16) as h:
        h.set_option("fanout_min", 0)
        want = h.forward(crops)
        for inflight, chunk, stage, depth in ((1, 64, 0, 2), (3, 64, 0, 2), (3, 64, 1, 2), (2, 100, 0, 1), (4, 33, 1, 4), (1, 128, 1, 2), (2, 128, -1, 2),
                                              (1, 64, 2, 2), (2, 128, 2, 2), (3, 40, 2, 1), (4, 33, 2, 4)):
            h.set_option("inflight", inflight)
            h.set_option("fanout_min", 128)
            h.set_option("fanout_chunk", chunk)
            h.set_option("fanout_stage", stage)
            h.set_option("fanout_depth", depth)
            for _ in range(7 if stage < 0 else 1):                     # (calibration: both forms warmed up, two timed calls each, then the faster)
                got = h.forward(crops)
                for a, b in zip(got, want):
                    assert np.array_equal(a, b), (inflight, chunk, stage, depth)
            y, am, lg = h.forward(crops[:130], want_logits=False)
            assert lg is None and np.array_equal(y, want[0][:130]) and np.array_equal(am, want[1][:130])
        # below the threshold nothing changes; submissions still work next to it
        assert np.array_equal(h.forward(crops[:100])[0], want[0][:100])
        t = h.submit(crops[:9])
        assert np.array_equal(h.forward(crops[:150])[0], want[0][:150])        # (a pending submission: the plain forward)
        assert np.array_equal(h.collect(t, 9)[0], want[0][:9])
        with pytest.raises(ValueError):
            h.set_option("fanout_depth", 9)
        # the fan-out's own engines (round 6): "inflight" stays what the caller set, whatever large calls came before
        h.set_option("inflight", 1)
        h.set_option("fanout_stage", 2)
        for engines in (1, 3, 2):
            h.set_option("fanout_engines", engines)
            got = h.forward(crops)
            for a, b in zip(got, want):
                assert np.array_equal(a, b), engines
        with pytest.raises(ValueError):
            h.set_option("fanout_engines", 5)
        with pytest.raises(ValueError):
            h.set_option("fanout_stage", 4)
