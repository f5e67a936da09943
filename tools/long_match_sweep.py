#!/usr/bin/env python3
"""Size per named case against the reference and speed per content kind for a list of
long-match settings "deep:fan[:chain]" (environment SETTINGS, space separated)."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as zo  # noqa: E402
import bench  # noqa: E402
import zlib_es_amd as z  # noqa: E402

golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
cases = [c for c in golden["cases"] if c["n"] >= 16384]
datas = [zo.gen(c["spec"]) for c in cases]
names = ["%s%d" % (c["spec"]["gen"][:6], c["n"]) + (str(c["spec"].get("k", ""))) for c in cases]
settings = os.environ.get("SETTINGS", "0:0 32:3").split()
NU = int(os.environ.get("NU", "2048"))
dev = torch.device("cuda")
full = bench.gen_units(20 * 512, 65536, 7, dev).view(-1, 65536)
idx = torch.arange(full.shape[0], device=dev) % 20
kinds = {"text": (0, 10), "binary": (10, 15), "random": (15, 18), "lowent": (18, 20)}


def speed(rows, ctype):
    n = rows.shape[0]
    in_off = np.arange(n + 1, dtype=np.uint64) * np.uint64(65536)
    plan = z.Plan(in_off, ctype, 0, 1)
    out = torch.empty(plan.out_bound + 64, dtype=torch.uint8, device=dev)
    flat = rows.reshape(-1)
    plan.run(flat.data_ptr(), out.data_ptr(), out.numel(), None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.run(flat.data_ptr(), out.data_ptr(), out.numel(), None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    off, _, _ = plan.fetch()
    plan.close()
    return n * 65536 / dt / 1e9, int(off[-1]) / (n * 65536)


for st in settings:
    parts = [int(x) for x in st.split(":")]
    z._native.set_long_match(parts[0], parts[1])
    z.set_search(parts[2] if len(parts) > 2 else 8, 32)
    line = []
    for ct in (1, 2):
        outs, _, _ = z.deflate_batch(datas, ct, 0)
        for d, o in zip(datas, outs):
            assert zo.inflate_raw(o, d.size) == d.tobytes()
        r = [len(o) / c["raw"][str(ct)]["len"] for o, c in zip(outs, cases)]
        line.append(" ".join("%s=%.3f" % (nm, v) for nm, v in zip(names, r)))
    print("== %s\n FX %s\n DY %s" % (st, line[0], line[1]), flush=True)
    sp = []
    for name, (a, b) in kinds.items():
        rows = full[(idx >= a) & (idx < b)][:NU].contiguous()
        g, r = speed(rows, 2)
        sp.append("%s %.1f GB/s (%.4f)" % (name, g, r))
    g, r = speed(full[:8192].contiguous(), 2)
    sp.append("mix %.1f GB/s (%.4f)" % (g, r))
    print(" " + " | ".join(sp), flush=True)
