#!/usr/bin/env python3
"""The device-resident path on real bytes instead of the bench's synthetic mix: the first GIB GiB of the files
under a directory (argument) (default: the ROCm libraries of the image, GPU code objects and x86 code), 64 KiB units,
DYNAMIC and FIXED, every unit verified by the GPU inflate verifier.  Prints speed, size and what the
dense-class pass took (by running once more with the pass off)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gpu_verify as gv  # noqa: E402
import zlib_es_amd as z  # noqa: E402

root = sys.argv[1] if len(sys.argv) > 1 else "/opt/rocm/lib"
limit = int(float(os.environ.get("GIB", "2")) * (1 << 30))
ub = 65536
buf = np.empty(limit, dtype=np.uint8)
have = seen = 0
for d, dirs, files in os.walk(root):  # (walked as it goes: a recursive glob of a large tree prints nothing for minutes)
    dirs.sort()
    for name in sorted(files):
        if have >= limit:
            break
        f = os.path.join(d, name)
        seen += 1
        if seen % 2000 == 0:
            print("%d files, %.2f GiB" % (seen, have / (1 << 30)), flush=True)
        if os.path.isfile(f) and not os.path.islink(f):
            try:
                b = np.fromfile(f, dtype=np.uint8, count=limit - have)
            except OSError:
                continue
            buf[have:have + b.size] = b
            have += b.size
    if have >= limit:
        break
nu = have // ub
dev = torch.device("cuda")
data = torch.from_numpy(buf[: nu * ub]).to(dev)
off = np.arange(nu + 1, dtype=np.uint64) * np.uint64(ub)
print("%d units of 64 KiB from %s" % (nu, root), flush=True)
for ctype in (2, 1):
    sizes = {}
    for dense in (1, 0):
        z._native.set_dense_search(z._native.DEFAULT_DENSE_CHAIN if dense else 0)
        plan = z.Plan(off, ctype, 0, 1)
        out = torch.empty(plan.out_bound + 64, dtype=torch.uint8, device=dev)
        plan.run(data.data_ptr(), out.data_ptr(), out.numel(), None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.run(data.data_ptr(), out.data_ptr(), out.numel(), None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        o, _, _ = plan.fetch()
        st = gv.verify(out, torch.from_numpy(o.astype(np.int64)).to(dev), data, torch.from_numpy(off.astype(np.int64)).to(dev),
                       final_each=True, primed=False)
        bad = int((st != 0).sum())
        sizes[dense] = np.diff(o.astype(np.int64))
        print("ctype %d dense pass %s: %.2f GB/s (%.1f ms), out/in %.4f, %d of %d units failed the verifier" % (
            ctype, "on" if dense else "off", nu * ub / dt / 1e9, dt * 1e3, int(o[-1]) / (nu * ub), bad, nu), flush=True)
        plan.close()
        del out
    z._native.set_dense_search()
    if "DUMP" in os.environ and ctype == 1:  # the units the pass took: sample statistics and what it bought
        host = buf[: nu * ub].reshape(nu, ub)
        for i in np.nonzero(sizes[1] != sizes[0])[0][: int(os.environ["DUMP"])]:
            d = host[i]
            stride = (ub - 65) // 31
            idx = np.arange(32)[:, None] * stride + np.arange(64)[None, :]
            b0, b1 = d[idx], d[idx + 1]
            one = (b0 == b1).all(axis=1)
            bb = b0[~one].reshape(-1)
            cnt = np.bincount(bb, minlength=256).astype(np.int64)
            print("unit %6d values %3d top %.2f squares %.3f equal %.2f one-value %.2f size %6d -> %6d (%.3f)" % (
                i, int((cnt > 0).sum()), cnt.max() / max(1, bb.size), (cnt * cnt).sum() / max(1, bb.size) ** 2,
                (b0[~one] == b1[~one]).mean() if bb.size else 0, one.mean(), sizes[0][i], sizes[1][i], sizes[1][i] / sizes[0][i]), flush=True)
    taken = int((sizes[1] != sizes[0]).sum())
    print("ctype %d: the pass took %d of %d units; size with / without %.4f" % (ctype, taken, nu, sizes[1].sum() / sizes[0].sum()), flush=True)
