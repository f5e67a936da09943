#!/usr/bin/env python3
"""Stress of the search kernel's wave-to-wave hand-offs: batches of units of uneven sizes are
compressed over and over with search settings that make the search itself short (so the
waves spend their time in the linking and parsing hand-offs), and every output stream is
checked on the device each time.

usage: stress_handoff.py [REPS] [UNITS] [small]     (small: units of 0 .. 2000 bytes)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import _gpu_verify as gv  # noqa: E402
import zlib_es_amd as z  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
nu = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
small = len(sys.argv) > 3 and sys.argv[3] == "small"
dev = torch.device("cuda:0")
torch.cuda.init()
data = bench.gen_units(nu, 65536, 5, dev)
rng = np.random.default_rng(3)
# uneven unit sizes over the same bytes: 1 B .. 400 KB
cuts = [0]
total = nu * 65536
while cuts[-1] < total:
    r = rng.random()
    if small:
        cuts.append(min(total, cuts[-1] + int(rng.integers(0, 2000))))
        continue
    n = int(rng.integers(1, 3000)) if r < 0.2 else int(rng.integers(3000, 140000)) if r < 0.9 else int(rng.integers(140000, 400000))
    cuts.append(min(total, cuts[-1] + n))
in_off = np.array(cuts, dtype=np.uint64)
d_in_off = torch.from_numpy(in_off.astype(np.int64)).to(dev)
bad_total = 0
t0 = time.time()
for rep in range(reps):
    N = z._native
    chain, nice, ctype, lazy, flags = [
        (1, 8, 1, 0, N.F_FINAL_EACH), (1, 3, 2, 0, N.F_FINAL_EACH), (2, 32, 2, 0, N.F_FINAL_EACH),
        (8, 32, 2, 0, N.F_FINAL_EACH), (8, 258, 2, 0, N.F_FINAL_EACH), (3, 32, 1, 8, N.F_FINAL_EACH),
        (8, 32, 2, 0, N.F_SYNC_ALIGN | N.F_FINAL_LAST | N.F_PRIME_WINDOW), (1, 258, 2, 4, N.F_SYNC_ALIGN | N.F_FINAL_LAST | N.F_PRIME_WINDOW),
    ][rep % 8]
    z.set_search(chain, nice)
    plan = z.Plan(in_off, ctype, lazy, flags)
    out = torch.empty(plan.out_bound + 64, dtype=torch.uint8, device=dev)
    plan.run(data.data_ptr(), out.data_ptr(), out.numel(), None)
    off, _, _ = plan.fetch()
    st = gv.verify(out, torch.from_numpy(off.astype(np.int64)).to(dev), data, d_in_off,
                   final_each=bool(flags & N.F_FINAL_EACH), primed=bool(flags & N.F_PRIME_WINDOW))
    bad = (st != 0).nonzero().flatten().tolist()
    bad_total += len(bad)
    if bad or rep % 10 == 0:
        print("rep %d chain %d nice %d ctype %d lazy %d flags %d: %d units, bad %s (%.0f s)" % (
            rep, chain, nice, ctype, lazy, flags, in_off.size - 1, [(u, int(in_off[u + 1] - in_off[u])) for u in bad[:5]],
            time.time() - t0), flush=True)
    del plan, out
z.set_search()
print("failures:", bad_total)
sys.exit(1 if bad_total else 0)
