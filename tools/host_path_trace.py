#!/usr/bin/env python3
"""Three zgpu_deflate_batch calls on 1 GiB in page-locked buffers, for a rocprofv3 kernel + memory-copy trace."""
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

N = z._native
L = N.lib()
nu, ub = 16384, 65536
src = bench.gen_units(nu, ub, 1, torch.device("cuda")).cpu().numpy()
off = (np.arange(nu + 1, dtype=np.uint64) * np.uint64(ub))
cap = int(N.deflate_bound(ub) + 48) * nu + 64
out_off = np.zeros(nu + 1, dtype=np.uint64)
pin_in, pin_out = N.HostBuffer(src.size), N.HostBuffer(cap)
pin_in.array[:] = src
for i in range(3):
    t0 = time.perf_counter()
    rc = L.zgpu_deflate_batch(pin_in.ptr, off.ctypes.data, nu, 2, 0, 1, pin_out.ptr, cap, out_off.ctypes.data, None, None)
    print("call %d rc %d %.2f ms" % (i, rc, (time.perf_counter() - t0) * 1e3), flush=True)
