#!/usr/bin/env python3
"""Rate of the host-buffer entry point zgpu_deflate_batch (PCIe included) on 1 GiB of the bench mix."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import zlib_es_amd as z  # noqa: E402

dev = torch.device("cuda")
nu, ub = 16384, 65536
data = bench.gen_units(nu, ub, 1, dev).cpu().numpy()
off = (np.arange(nu + 1, dtype=np.uint64) * np.uint64(ub))
cap = int(z._native.deflate_bound(ub) + 48) * nu + 64
out = np.empty(cap, dtype=np.uint8)
out_off = np.zeros(nu + 1, dtype=np.uint64)
crc = np.zeros(nu, dtype=np.uint32)
L = z._native.lib()
best = 0
for it in range(5):
    t0 = time.perf_counter()
    rc = L.zgpu_deflate_batch(data.ctypes.data, off.ctypes.data, nu, 2, 0, 1, out.ctypes.data, cap, out_off.ctypes.data, None, None)
    dt = time.perf_counter() - t0
    best = max(best, nu * ub / dt / 1e9)
    print("C ABI host path rc=%d: %.2f GB/s (%.3f s)" % (rc, nu * ub / dt / 1e9, dt), flush=True)
t0 = time.perf_counter()
rc = L.zgpu_deflate_batch(data.ctypes.data, off.ctypes.data, nu, 2, 0, 1, out.ctypes.data, cap, out_off.ctypes.data, crc.ctypes.data, None)
print("with per-unit CRC-32 rc=%d: %.2f GB/s" % (rc, nu * ub / (time.perf_counter() - t0) / 1e9))
t0 = time.perf_counter()
c = z._native.checksum(data)
print("zgpu_checksum of 1 GiB host: %.2f GB/s" % (nu * ub / (time.perf_counter() - t0) / 1e9))
print("BEST %.2f" % best)
