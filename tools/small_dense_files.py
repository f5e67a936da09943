#!/usr/bin/env python3
"""Files of 300 bytes to 64 KiB of two to six equally likely byte values, eight of each in one zgpu_deflate_batch
call: size against the reference's algorithm (C restatement) under FIXED and DYNAMIC at the default settings, and
the time of the call."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import _oracle as zo
import zlib_es_amd as z
for k in (2, 3, 4, 5, 6):
    for n in (300, 1000, 3000, 8000, 20000, 65536):
        ds = [zo.gen({"gen": "small-alpha", "n": n, "seed": 100 * k + i, "k": k}) for i in range(8)]
        row = []
        for ct in (1, 2):
            outs, _, _ = z.deflate_batch(ds, ct, 0)
            t0 = time.perf_counter(); outs, _, _ = z.deflate_batch(ds, ct, 0); dt = time.perf_counter() - t0
            g = sum(len(o) for o in outs); r = sum(len(zo.rawdeflate(d, ct)) for d in ds)
            row.append(g / r)
        print("k=%d n=%5d FIXED %.3f DYNAMIC %.3f  %.2f ms" % (k, n, row[0], row[1], dt * 1e3), flush=True)
