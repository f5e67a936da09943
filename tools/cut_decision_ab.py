#!/usr/bin/env python3
"""Batches of few large units with the library's own decision about cutting them into pieces (split = 1 MiB, the
default limit) and with cutting switched off (split = 0): GB/s, search and emit kernel times, out/in."""
import sys, time
import numpy as np, torch
torch.cuda.init()
sys.path.insert(0, '/root/repo')
import bench, zlib_es_amd as z
dev = torch.device("cuda")
data = bench.gen_units(16384, 65536, 7, dev).view(-1)
for nunits, ub in ((1024, 1 << 20), (512, 1 << 20), (600, 1 << 20), (256, 4 << 20)):
    for split in (1 << 20, 0):
        z._native.set_split(split)
        in_off = (np.arange(nunits + 1, dtype=np.uint64) * np.uint64(ub))
        plan = z.Plan(in_off, 2, 0, 1)
        out = torch.empty(plan.out_bound + 64, dtype=torch.uint8, device=dev)
        d = data[: nunits * ub]
        for _ in range(2): plan.run(d.data_ptr(), out.data_ptr(), out.numel(), None)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(4): plan.run(d.data_ptr(), out.data_ptr(), out.numel(), None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 4
        s, e = plan.phase_ms()
        off, _, _ = plan.fetch()
        print("%d x %d KiB split=%d: %.2f GB/s (%.2f ms; search %.2f emit %.2f) out/in %.5f" % (nunits, ub >> 10, split, nunits * ub / dt / 1e9, dt * 1e3, s, e, int(off[-1]) / (nunits * ub)), flush=True)
        plan.close()
