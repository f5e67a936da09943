import time, sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import bench, zlib_es_amd as z
dev = torch.device("cuda")
nu, ub = 16384, 65536
data = bench.gen_units(nu, ub, 1, dev).cpu().numpy()
units = [data[i*ub:(i+1)*ub] for i in range(nu)]
for it in range(3):
    t0 = time.perf_counter()
    outs, _, _ = z.deflate_batch(units, 2, 0)
    dt = time.perf_counter() - t0
    print("host-buffer path: %.2f GB/s (%.3f s for %d MiB), out %d MiB" % (nu*ub/dt/1e9, dt, nu*ub>>20, sum(len(o) for o in outs)>>20), flush=True)
# raw C call without the python list handling
off = (np.arange(nu + 1, dtype=np.uint64) * np.uint64(ub))
cap = int(z._native.deflate_bound(ub) + 48) * nu + 64
out = np.empty(cap, dtype=np.uint8); out_off = np.zeros(nu + 1, dtype=np.uint64)
L = z._native.lib()
for it in range(3):
    t0 = time.perf_counter()
    rc = L.zgpu_deflate_batch(data.ctypes.data, off.ctypes.data, nu, 2, 0, 1, out.ctypes.data, cap, out_off.ctypes.data, None, None)
    dt = time.perf_counter() - t0
    print("C ABI host path rc=%d: %.2f GB/s (%.3f s)" % (rc, nu*ub/dt/1e9, dt), flush=True)
