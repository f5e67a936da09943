#!/usr/bin/env python3
"""Small alphabets (k distinct byte values, uniform): size against the reference's algorithm
and speed of NU units of 64 KiB, for the default search and for search settings "chain:nice"
(environment SETTINGS, space separated; nice > 32 selects the thorough kernel)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as zo  # noqa: E402
import zlib_es_amd as z  # noqa: E402

KS = [int(k) for k in os.environ.get("KS", "2 3 4 5 6 8 12 16").split()]
SETTINGS = os.environ.get("SETTINGS", "8:32 128:258 1024:258 32768:258").split()
NU = int(os.environ.get("NU", "64"))
UB = 65536
dev = torch.device("cuda")
if "MINLAZY" in os.environ:
    z._native.set_min_lazy(int(os.environ["MINLAZY"]))


def speed(units, ctype):
    flat = torch.from_numpy(np.concatenate(units)).to(dev)
    in_off = np.arange(len(units) + 1, dtype=np.uint64) * np.uint64(UB)
    plan = z.Plan(in_off, ctype, 0, 1)
    out = torch.empty(plan.out_bound + 64, dtype=torch.uint8, device=dev)
    plan.run(flat.data_ptr(), out.data_ptr(), out.numel(), None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.run(flat.data_ptr(), out.data_ptr(), out.numel(), None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    plan.close()
    return len(units) * UB / dt / 1e9, dt * 1e3


for k in KS:
    units = [zo.gen({"gen": "small-alpha", "n": UB, "seed": 100 * k + i, "k": k}) for i in range(NU)]
    ref = {ct: [len(zo.rawdeflate(u, ct)) for u in units[:2]] for ct in (1, 2)}
    for st in SETTINGS:
        c, nice = [int(x) for x in st.split(":")]
        z.set_search(c, nice)
        r = {}
        for ct in (1, 2):
            outs, _, _ = z.deflate_batch(units[:2], ct, 0)
            for d, o in zip(units[:2], outs):
                assert zo.inflate_raw(o, d.size) == d.tobytes()
            r[ct] = max(len(o) / s for o, s in zip(outs, ref[ct]))
        g, ms = speed(units, 2)
        print("k=%-3d %-10s FIXED %.3f DYNAMIC %.3f | %d units: %.2f GB/s (%.1f ms)" % (k, st, r[1], r[2], NU, g, ms),
              flush=True)
    z.set_search()
