import time, numpy as np, torch, ctypes
rt = torch.cuda.cudart()
for mb, n in ((9.6, 64), (77, 512)):
    a = np.random.randint(0, 255, size=(n, 224, 224, 3), dtype=np.uint8)
    d = torch.empty(a.nbytes, dtype=torch.uint8, device="cuda")
    t = torch.from_numpy(a.reshape(-1))
    torch.cuda.synchronize()
    # pageable copy
    ts = []
    for i in range(8):
        t0 = time.perf_counter(); d.copy_(t); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    pg = np.median(ts[2:])
    regs, unregs, cps = [], [], []
    for i in range(6):
        t0 = time.perf_counter(); r = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0); t1 = time.perf_counter()
        assert int(r) == 0, r
        d.copy_(t, non_blocking=True); torch.cuda.synchronize(); t2 = time.perf_counter()
        r = rt.cudaHostUnregister(a.ctypes.data); t3 = time.perf_counter()
        regs.append(t1 - t0); cps.append(t2 - t1); unregs.append(t3 - t2)
    print(f"{n} crops ({a.nbytes/1e6:.1f} MB): pageable H2D {pg*1e3:.3f} ms ({a.nbytes/pg/1e9:.1f} GB/s) | register {np.median(regs[1:])*1e3:.3f} ms, copy from registered {np.median(cps[1:])*1e3:.3f} ms ({a.nbytes/np.median(cps[1:])/1e9:.1f} GB/s), unregister {np.median(unregs[1:])*1e3:.3f} ms", flush=True)
    # fresh buffer each time (first-touch registration)
    regs = []
    for i in range(4):
        b = np.empty_like(a); b[:] = a
        t0 = time.perf_counter(); r = rt.cudaHostRegister(b.ctypes.data, b.nbytes, 0); t1 = time.perf_counter()
        rt.cudaHostUnregister(b.ctypes.data)
        regs.append(t1 - t0)
    print(f"   fresh buffers: register {np.median(regs)*1e3:.3f} ms")
