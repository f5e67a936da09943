#!/usr/bin/env python3
"""Host <-> device copy rates on this box (pageable and pinned), to place the host-buffer entry
points' PCIe-inclusive rate."""
import time

import numpy as np
import torch

n = 1 << 30
dev = torch.device("cuda:0")
torch.cuda.init()
h = np.random.default_rng(0).integers(0, 256, n, dtype=np.uint8)
d = torch.empty(n, dtype=torch.uint8, device=dev)
ht = torch.from_numpy(h)
for name, src in (("pageable", ht), ("pinned", ht.pin_memory())):
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d.copy_(src, non_blocking=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        src.copy_(d)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print("%s: H2D %.1f GB/s, D2H %.1f GB/s" % (name, n / (t1 - t0) / 1e9, n / (t2 - t1) / 1e9), flush=True)
t0 = time.perf_counter()
p = ht.pin_memory()
print("pin_memory (alloc + copy) of 1 GiB: %.3f s" % (time.perf_counter() - t0))
t0 = time.perf_counter()
h2 = h.copy()
print("host memcpy of 1 GiB, one thread: %.1f GB/s" % (n / (time.perf_counter() - t0) / 1e9))
# pinning the caller's pages in place (hipHostRegister) instead of staging them
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
for sz in (64 << 20, 1 << 30):
    t0 = time.perf_counter()
    rc = hip.hipHostRegister(h.ctypes.data, sz, 0)
    t1 = time.perf_counter()
    rc2 = hip.hipMemcpy(d.data_ptr(), h.ctypes.data, sz, 1)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hip.hipHostUnregister(h.ctypes.data)
    t3 = time.perf_counter()
    print("hipHostRegister %d MiB: rc %d, %.1f ms (%.1f GB/s); H2D from registered %.1f GB/s; unregister %.1f ms" % (
        sz >> 20, rc, (t1 - t0) * 1e3, sz / (t1 - t0) / 1e9, sz / (t2 - t1) / 1e9, (t3 - t2) * 1e3))
# threads copying side by side
import threading
for nt in (2, 4, 8):
    dst = np.empty_like(h)
    parts = np.array_split(np.arange(n // 4096), nt)
    def work(i):
        a, b = int(parts[i][0]) * 4096, (int(parts[i][-1]) + 1) * 4096
        np.copyto(dst[a:b], h[a:b])
    ths = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    print("host memcpy of 1 GiB, %d threads: %.1f GB/s" % (nt, n / (time.perf_counter() - t0) / 1e9))
