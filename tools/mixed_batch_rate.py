#!/usr/bin/env python3
"""Rate of a zip-like batch of mixed sizes on the GPU (device-resident, DYNAMIC, CRC-32 per file): 200,000 files
of 1-4 KiB and 64 files of 1 MiB in random order.  The small files run in the small-window search variants and the
wave-per-unit emit kernel, the large ones are cut into pieces that fill the GPU."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
torch.cuda.init()
import bench, zlib_es_amd as z
dev = torch.device("cuda")
# a zip-like batch: 200k files of 1-4 KiB and 64 files of 1 MiB, mixed content
rng = np.random.default_rng(3)
sizes = np.concatenate([rng.integers(1024, 4096, 200000), np.full(64, 1 << 20)])
rng.shuffle(sizes)
total = int(sizes.sum())
data = bench.gen_units(total // 65536 + 1, 65536, 7, dev).view(-1)[:total].contiguous()
in_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
plan = z.Plan(in_off, 2, 0, z._native.F_FINAL_EACH | z._native.F_WANT_CRC)
out = torch.empty(plan.out_bound + 64, dtype=torch.uint8, device=dev)
for _ in range(2):
    plan.run(data.data_ptr(), out.data_ptr(), out.numel(), None)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    plan.run(data.data_ptr(), out.data_ptr(), out.numel(), None)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 3
off, crc, _ = plan.fetch(want_crc=True)
print("mixed batch %d units %.1f MiB: %.2f GB/s (%.2f ms), out/in %.4f" % (len(sizes), total / 2**20, total / dt / 1e9, dt * 1e3, int(off[-1]) / total))
