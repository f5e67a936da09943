#!/usr/bin/env python3
"""Device-resident rate of zgpu_plan_run over batches of unusual shapes (sanity sweep: looks for shapes that
fall off the fast paths).  Prints GB/s, ms and out/in per shape."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import zlib_es_amd as z  # noqa: E402

dev = torch.device("cuda")
torch.cuda.init()
POOL = bench.gen_units(8192, 65536, 7, dev).view(-1)  # 512 MiB of the mix


def run(name, sizes, ctype=2, flags=1, shift=0):
    sizes = np.asarray(sizes, dtype=np.uint64)
    total = int(sizes.sum())
    data = POOL[shift:shift + total]
    if data.numel() < total:
        data = POOL.repeat((total + shift) // POOL.numel() + 1)[shift:shift + total]
    data = data.contiguous() if shift == 0 else data  # a shifted view keeps its odd alignment
    in_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    plan = z.Plan(in_off, ctype, 0, flags)
    out = torch.empty(plan.out_bound + 64, dtype=torch.uint8, device=dev)
    for _ in range(2):
        plan.run(data.data_ptr(), out.data_ptr(), out.numel(), None)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 3
    for _ in range(reps):
        plan.run(data.data_ptr(), out.data_ptr(), out.numel(), None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    off, _, _ = plan.fetch()
    print("%-44s %8d units %8.1f MiB %7.2f GB/s %8.2f ms  out/in %.4f" % (name, len(sizes), total / 2**20, total / dt / 1e9, dt * 1e3,
                                                                     int(off[-1]) / max(total, 1)), flush=True)
    plan.close()


rng = np.random.default_rng(1)
run("64 KiB units", [65536] * 8192)
run("64 KiB units, input shifted by 1 byte", [65536] * 8191, shift=1)
run("65537-byte units (a 1-byte second block)", [65537] * 8000)
run("100-byte units", [100] * 1000000)
run("sizes 0..200 bytes", rng.integers(0, 201, 1000000))
run("17..40 KiB units", rng.integers(17000, 40000, 16000))
run("1 MiB units", [1 << 20] * 512)
run("one 256 MiB unit", [256 << 20])
run("three units of 100 MiB", [100 << 20] * 3)
run("64 KiB units, FIXED", [65536] * 8192, ctype=1)
run("64 KiB units, stored", [65536] * 8192, ctype=0)
run("4 KiB units, FIXED", [4096] * 131072, ctype=1)
run("log-uniform 64 B .. 4 MiB", np.exp(rng.uniform(np.log(64), np.log(4 << 20), 3000)).astype(np.int64))
