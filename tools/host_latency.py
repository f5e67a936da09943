#!/usr/bin/env python3
"""Latency of single host-buffer calls (what one RawDeflate#compress() costs through the C ABI)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as zo
import zlib_es_amd as z
for n in (100, 4096, 65536, 1 << 20, 16 << 20):
    d = zo.gen({"gen": "text", "n": n, "seed": 1})
    z.deflate_batch([d], 2, 0)
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps):
        z.deflate_batch([d], 2, 0)
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        z.checksum(d)
    dc = (time.perf_counter() - t0) / reps
    print("n=%9d  deflate_batch %.3f ms (%.2f GB/s)   checksum %.3f ms" % (n, dt * 1e3, n / dt / 1e9, dc * 1e3), flush=True)
