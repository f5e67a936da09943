#!/usr/bin/env python3
"""Rate of zgpu_deflate_batch on 1 GiB of the bench mix with the caller's buffers in pageable memory,
in page-locked memory (zgpu_host_alloc) and in pageable memory registered for the call
(hipHostRegister / hipHostUnregister timed with it)."""
import ctypes
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
dev = torch.device("cuda")
nu, ub = int(os.environ.get("NU", 16384)), 65536
src = bench.gen_units(nu, ub, 1, dev).cpu().numpy()
off = (np.arange(nu + 1, dtype=np.uint64) * np.uint64(ub))
cap = int(N.deflate_bound(ub) + 48) * nu + 64
out_off = np.zeros(nu + 1, dtype=np.uint64)
hip = ctypes.CDLL("libamdhip64.so")


def rate(in_ptr, out_ptr, label, pre=None, post=None, two_step=False):
    best = 0.0
    for _ in range(5):
        t0 = time.perf_counter()
        if pre:
            pre()
        if two_step:
            job = ctypes.c_void_p()
            total = ctypes.c_uint64()
            rc = L.zgpu_deflate_batch_begin(in_ptr, off.ctypes.data, nu, 2, 0, 1, ctypes.byref(job), ctypes.byref(total))
            assert rc == 0, rc
            rc = L.zgpu_deflate_batch_finish(job, out_ptr, cap, out_off.ctypes.data, None, None)
        else:
            rc = L.zgpu_deflate_batch(in_ptr, off.ctypes.data, nu, 2, 0, 1, out_ptr, cap, out_off.ctypes.data, None, None)
        if post:
            post()
        dt = time.perf_counter() - t0
        assert rc == 0, rc
        best = max(best, nu * ub / dt / 1e9)
    print("%-58s %6.2f GB/s" % (label, best), flush=True)
    return best


L.zgpu_deflate_batch_begin.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                                                               ctypes.c_void_p, ctypes.c_void_p]
L.zgpu_deflate_batch_finish.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p]
out_pg = np.empty(cap, dtype=np.uint8)
out_pg[::4096] = 0
rate(src.ctypes.data, out_pg.ctypes.data, "pageable in, pageable out")
pin_in, pin_out = N.HostBuffer(src.size), N.HostBuffer(cap)
pin_in.array[:] = src
rate(pin_in.ptr, out_pg.ctypes.data, "page-locked in, pageable out")
rate(pin_in.ptr, pin_out.ptr, "page-locked in, page-locked out")
rate(src.ctypes.data, pin_out.ptr, "pageable in, page-locked out")
rate(pin_in.ptr, pin_out.ptr, "page-locked in and out, two-step call", two_step=True)
rate(src.ctypes.data, out_pg.ctypes.data, "pageable in and out, two-step call", two_step=True)
ref = bytes(out_pg[: int(out_off[-1])])
assert bytes(pin_out.array[: int(out_off[-1])]) == ref, "same bytes whatever the memory"
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
rate(src.ctypes.data, pin_out.ptr, "pageable in registered per call, page-locked out",
     pre=lambda: hip.hipHostRegister(src.ctypes.data, src.size, 0), post=lambda: hip.hipHostUnregister(src.ctypes.data))
