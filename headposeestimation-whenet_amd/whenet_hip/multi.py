"""One process, several GPUs: the batch of ONE get_angle call split over the devices of this node.

SURVEY.md §8e: crops are independent, so the path shards as a contiguous batch split with replicated weights and no
data-path collective.  whenet_hip/shard.py is the one-process-per-GPU form (torch.distributed / RCCL, what bench.py
--gpus N runs); this module is the form the reference's own call shape needs -- `WHENet(snapshot).get_angle(crops)`
from a single Python process (/root/reference/whenet.py:22, demo_video.py:27): one libwhenet_hip handle per device,
each driven by its OWN feeder thread (a handle is not thread-safe; ctypes releases the GIL for the duration of the C
call, so the devices really run concurrently), shards by shard_bounds(), results returned in crop order.  The kernels
are batch-invariant, so the result is bitwise that of one handle.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Optional, Sequence

import numpy as np

from .shard import shard_bounds


class MultiDeviceHandle:
    """Same surface as whenet_hip._lib.Handle for what whenet.WHENet uses (forward, forward_f32, info, set_option,
    close), over one handle per entry of `devices` (a device may be listed more than once: that many handles on it)."""

    def __init__(self, snapshot, devices: Sequence[int], dtype: int, handle_factory: Optional[Callable] = None,
                 min_shard: int = 1):
        if len(devices) < 1:
            raise ValueError("devices must name at least one GPU")
        if handle_factory is None:
            from . import _lib
            handle_factory = _lib.Handle
        self.devices = [int(d) for d in devices]
        self.dtype = dtype
        self.min_shard = int(min_shard)
        # each handle is created, used and destroyed on its own thread: exactly one thread ever touches it
        self._pools: List[ThreadPoolExecutor] = [ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"whenet-gpu{d}")
                                                 for d in self.devices]
        self._handles = []
        futs = [p.submit(handle_factory, snapshot, device=d, dtype=dtype) for p, d in zip(self._pools, self.devices)]
        made, err = [], None
        for f in futs:                              # every future is awaited: a handle created AFTER the first failure is still ours
            try:
                made.append(f.result())
            except BaseException as e:              # noqa: BLE001
                made.append(None)
                err = err or e
        if err is not None:
            for p, h in zip(self._pools, made):     # each created handle is closed on the thread that made it
                if h is not None:
                    try:
                        p.submit(h.close).result()
                    except Exception:               # noqa: BLE001
                        pass
            for p in self._pools:
                p.shutdown(wait=True)
            self._pools = []
            raise err
        self._handles = made
        self.device = self.devices[0]

    # ---- plumbing ---------------------------------------------------------------------------------------------------
    def _each(self, fn):
        futs = [p.submit(fn, h) for p, h in zip(self._pools, self._handles)]
        return [f.result() for f in futs]

    def _shards(self, n: int):
        """[(handle index, lo, hi)] of the non-empty shards: as many devices as have >= min_shard crops to do."""
        world = max(1, min(len(self._handles), n // max(1, self.min_shard))) if n > 0 else 1
        out = []
        for r in range(world):
            lo, hi = shard_bounds(n, world, r)
            if hi > lo:
                out.append((r, lo, hi))
        return out

    def _split_run(self, x: np.ndarray, method: str, want_logits: bool):
        n = x.shape[0]
        shards = self._shards(n)
        if len(shards) <= 1:
            r = self._pools[0].submit(getattr(self._handles[0], method), x, want_logits).result()
            return r
        futs = [(lo, hi, self._pools[i].submit(getattr(self._handles[i], method), x[lo:hi], want_logits)) for i, lo, hi in shards]
        ypr = np.empty((n, 3), np.float32)
        am = np.empty((n, 3), np.int32)
        lg = np.empty((n, 252), np.float32) if want_logits else None
        err = None
        for lo, hi, f in futs:                      # (every future is awaited even after a failure: no orphan work)
            try:
                y, a, l = f.result()
                ypr[lo:hi], am[lo:hi] = y, a
                if want_logits:
                    lg[lo:hi] = l
            except BaseException as e:              # noqa: BLE001
                err = err or e
        if err is not None:
            raise err
        return ypr, am, lg

    # ---- Handle surface ---------------------------------------------------------------------------------------------
    def forward(self, crops: np.ndarray, want_logits: bool = True):
        return self._split_run(crops, "forward", want_logits)

    def forward_f32(self, x: np.ndarray, want_logits: bool = True):
        return self._split_run(x, "forward_f32", want_logits)

    def set_option(self, key: str, value: int):
        self._each(lambda h: h.set_option(key, value))

    def info(self):
        return self._pools[0].submit(self._handles[0].info).result()

    def __getattr__(self, name):
        # single-stage / frame / detector entry points are not sharded: they go to the first device's handle, on its thread
        if name.startswith("_"):
            raise AttributeError(name)
        target = getattr(self._handles[0], name)
        if not callable(target):
            return target
        return lambda *a, **k: self._pools[0].submit(target, *a, **k).result()

    def close(self):
        for p, h in zip(self._pools, self._handles):
            try:
                p.submit(h.close).result()
            except Exception:                       # noqa: BLE001
                pass
        self._handles = []
        for p in self._pools:
            p.shutdown(wait=True)
        self._pools = []
