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
