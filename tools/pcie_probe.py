#!/usr/bin/env python3
"""What the host->device leg of whenet_forward_u8 costs on this box: pageable vs pinned copies of a 64-crop batch
(9.6 MB), chunked, and the price of hipHostRegister."""
import time, numpy as np, torch
dev = torch.device("cuda", 0)
N = 64 * 150528
a = np.random.default_rng(0).integers(0, 256, N, dtype=np.uint8)
d = torch.empty(N, dtype=torch.uint8, device=dev)
t = torch.from_numpy(a)
def tm(fn, it=30):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6
print(f"pageable 9.6 MB  H2D: {tm(lambda: d.copy_(t, non_blocking=True)):8.1f} us")
for k in (2, 4, 8):
    c = N // k
    print(f"pageable in {k} chunks : {tm(lambda: [d[i*c:(i+1)*c].copy_(t[i*c:(i+1)*c], non_blocking=True) for i in range(k)]):8.1f} us")
p = torch.empty(N, dtype=torch.uint8).pin_memory()
p.copy_(t)
print(f"pinned   9.6 MB  H2D: {tm(lambda: d.copy_(p, non_blocking=True)):8.1f} us")
print(f"host memcpy 9.6 MB (1 thread, pageable -> pinned): {tm(lambda: p.copy_(t)):8.1f} us")
rt = torch.cuda.cudart()
b = np.random.default_rng(1).integers(0, 256, N, dtype=np.uint8)
tb = torch.from_numpy(b)
t0 = time.perf_counter(); r = rt.cudaHostRegister(tb.data_ptr(), N, 0); t1 = time.perf_counter()
print(f"hipHostRegister 9.6 MB: {(t1 - t0) * 1e6:8.1f} us (rc {r})")
print(f"registered 9.6 MB H2D: {tm(lambda: d.copy_(tb, non_blocking=True)):8.1f} us")
t0 = time.perf_counter(); rt.cudaHostUnregister(tb.data_ptr()); t1 = time.perf_counter()
print(f"hipHostUnregister: {(t1 - t0) * 1e6:8.1f} us")
for _ in range(3):
    t0 = time.perf_counter(); rt.cudaHostRegister(tb.data_ptr(), N, 0); t1 = time.perf_counter(); rt.cudaHostUnregister(tb.data_ptr()); t2 = time.perf_counter()
    print(f"  again: register {(t1 - t0) * 1e6:8.1f} us, unregister {(t2 - t1) * 1e6:8.1f} us")
