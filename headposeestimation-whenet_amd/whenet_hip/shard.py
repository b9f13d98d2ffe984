"""Batch-shard WHENet across the GPUs of one node: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no multi-GPU code on this path.  Crops are
independent (inference-mode BN, no cross-sample op), so the path shards as an
embarrassingly-parallel contiguous batch split with NO collective on the data path:

    rank r owns crops [lo_r, hi_r)   (sizes differ by at most one),
    weights are replicated (17 MB f32 / 8.6 MB f16 per GPU).

The only collectives, both off the hot path and latency-bound:
  * broadcast of the packed snapshot from rank 0 at construction (so that only one rank
    needs the file);
  * an optional all_gather of the [n_r,3] angle / argmax shards so every rank (or the
    caller on rank 0) holds the full result.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of n items over `world` ranks; the first n % world ranks get one more."""
    if world < 1 or not (0 <= rank < world) or n < 0:
        raise ValueError(f"bad shard request n={n} world={world} rank={rank}")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _parse_cpulist(text: str):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of sysfs `local_cpulist` / `cpulist`)."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def gpu_numa_cpus(pci_bus_id: str, sysfs_root: str = "/sys"):
    """(numa_node, cpus) of the PCI device `pci_bus_id` ('0000:c1:00.0') as the kernel reports them:
    <sysfs>/bus/pci/devices/<id>/numa_node and the node's cpulist (falling back to the device's local_cpulist).
    numa_node = -1 (no affinity known, e.g. inside most VMs) gives (-1, [])."""
    import os
    base = os.path.join(sysfs_root, "bus", "pci", "devices", pci_bus_id.lower())
    try:
        with open(os.path.join(base, "numa_node")) as f:
            node = int(f.read().strip())
    except (OSError, ValueError):
        return -1, []
    if node < 0:
        return -1, []
    for path in (os.path.join(sysfs_root, "devices", "system", "node", f"node{node}", "cpulist"),
                 os.path.join(base, "local_cpulist")):
        try:
            with open(path) as f:
                cpus = _parse_cpulist(f.read())
            if cpus:
                return node, cpus
        except (OSError, ValueError):
            continue
    return node, []


def gpu_pci_bus_id(local_rank: int) -> str:
    """'dddd:bb:dd.0' of HIP device `local_rank` (torch's device properties)."""
    import torch
    p = torch.cuda.get_device_properties(local_rank)
    return f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"


def bind_rank_to_gpu_numa(pci_bus_id: str, index_on_node: int = 0, peers_on_node: int = 1,
                          sysfs_root: str = "/sys") -> dict:
    """Pin this process (one rank per GPU) to the CPUs of its GPU's NUMA node, so that the host thread that enqueues
    the launches and the pinned staging buffers it touches sit next to the GPU's PCIe root.  When `peers_on_node`
    ranks have their GPUs on the same node, the node's CPUs are divided between them and this rank takes slice
    `index_on_node`.  Returns what was done (for the bench line); never raises: without NUMA information (node -1,
    as inside most VMs) the affinity is left as it is."""
    import os
    info = {"pci_bus_id": pci_bus_id, "numa_node": -1, "bound": False, "cpus": None}
    try:
        node, cpus = gpu_numa_cpus(pci_bus_id, sysfs_root)
        info["numa_node"] = node
        if node < 0 or not cpus:
            return info
        allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        if not allowed:
            return info
        peers = max(1, int(peers_on_node))
        per = max(1, len(allowed) // peers)
        k = int(index_on_node) % peers
        mine = allowed[k * per:(k + 1) * per] or allowed
        os.sched_setaffinity(0, mine)
        info.update(bound=True, cpus=len(mine), first_cpu=mine[0], last_cpu=mine[-1])
    except (OSError, AttributeError, ValueError) as e:
        info["error"] = str(e)[:120]
    return info


def broadcast_bytes(blob: Optional[bytes], src: int = 0, device=None) -> bytes:
    """Broadcast a byte string from rank `src` to every rank of the default process group."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    dev = device if device is not None else "cpu"
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    if rank == src:
        buf = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src=src)
    return bytes(buf.cpu().numpy().tobytes())


class ShardedWHENet:
    """get_angle over a batch that is split across the ranks of the default process group.

    forward : callable(uint8 crops [m,224,224,3]) -> (ypr float32 [m,3], argmax int32 [m,3]);
              by default a libwhenet_hip handle on this rank's GPU.  (The CPU/gloo tests
              inject the oracle here; the product never does.)
    """

    def __init__(self, snapshot=None, *, dtype="f32", device: Optional[int] = None,
                 forward: Optional[Callable] = None, comm_device=None):
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.comm_device = comm_device
        self._model = None
        if forward is not None:
            self._forward = forward
            return
        import os
        from whenet import WHENet       # the drop-in module next to this package
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", self.rank))
        if self.world > 1:
            blob = None
            if self.rank == 0:
                from . import weights as W
                if snapshot is None:
                    blob = W.pack(W.synthetic(1234))
                elif isinstance(snapshot, (bytes, bytearray)):
                    blob = bytes(snapshot)
                else:
                    with open(snapshot, "rb") as f:
                        blob = f.read()
            snapshot = broadcast_bytes(blob, 0, comm_device)
        self._model = WHENet(snapshot, device=device, dtype=dtype)

        def fwd(u8):
            ypr, am, _ = self._model._forward(u8)
            return ypr, am
        self._forward = fwd

    def bounds(self, n: int) -> Tuple[int, int]:
        return shard_bounds(n, self.world, self.rank)

    def forward_local(self, crops: np.ndarray, global_batch: bool = True):
        """Run this rank's shard.  With global_batch=True `crops` is the whole batch (every
        rank passes the same array) and the shard is sliced here; otherwise it already is
        this rank's shard."""
        from ._lib import as_uint8_crops
        crops = np.asarray(crops)
        if crops.ndim != 4 or tuple(crops.shape[1:]) != (224, 224, 3):
            raise ValueError(f"Error when checking input: expected input to have shape "
                             f"(None, 224, 224, 3) but got array with shape {crops.shape}")
        if global_batch:
            lo, hi = self.bounds(crops.shape[0])
            crops = crops[lo:hi]
        if crops.shape[0] == 0:
            return np.empty((0, 3), np.float32), np.empty((0, 3), np.int32)
        # same validation / dtype handling as WHENet.get_angle (whenet.py:22-27): the C side reads
        # raw bytes, so nothing but a contiguous uint8 [m,224,224,3] array may reach it
        return self._forward(as_uint8_crops(crops))

    def get_angle(self, crops: np.ndarray):
        """Whole-batch result on every rank: shard, run, all_gather.  Returns (yaw, pitch, roll)
        like whenet.py:22-34, each float32 (N,), in the original crop order."""
        crops = np.asarray(crops)
        n = crops.shape[0] if crops.ndim == 4 else 0
        ypr, am = self.forward_local(crops, global_batch=True)
        if self.world == 1:
            return ypr[:, 0].copy(), ypr[:, 1].copy(), ypr[:, 2].copy()
        import torch
        sizes = [shard_bounds(n, self.world, r) for r in range(self.world)]
        mx = max(hi - lo for lo, hi in sizes)
        dev = self.comm_device if self.comm_device is not None else "cpu"
        mine = torch.zeros((mx, 3), dtype=torch.float32, device=dev)
        if ypr.shape[0]:
            mine[:ypr.shape[0]] = torch.from_numpy(ypr).to(dev)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        full = np.concatenate([p.cpu().numpy()[:hi - lo] for p, (lo, hi) in zip(parts, sizes)], axis=0)
        return full[:, 0].copy(), full[:, 1].copy(), full[:, 2].copy()

    predict = get_angle

    def close(self):
        if self._model is not None:
            self._model.close()
            self._model = None
