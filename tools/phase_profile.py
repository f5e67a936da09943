#!/usr/bin/env python3
"""Per-phase cycle breakdown of the DEFLATE kernel (diagnostic build
libzgpu_prof.so: `make -C zlib.es_amd/csrc prof`).  Prints, per content kind,
thread-0 cycles per input byte spent in each phase."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ZGPU_LIB"] = os.environ.get("ZGPU_PROF_LIB") or os.path.join(ROOT, "zlib.es_amd", "libzgpu_prof.so")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import zlib_es_amd as z  # noqa: E402

EPHASES = ["queue", "hist", "sort", "treeL", "treeD", "rle", "treeCL", "hdrbits", "tokens", "finish"]
PHASES = ["load", "blkwait", "deep", "fan", "horizon", "cert", "deepwalk", "emit-tok", "lag+ld", "sort", "linkwait", "link", "pass1", "pass2", "prep", "parsewait"]


def run(name, data, nu, ub, ctype, lazy):
    L = z._native.lib()
    L.zgpu_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    in_off = np.arange(nu + 1, dtype=np.uint64) * np.uint64(ub)
    plan = z.Plan(in_off, ctype, lazy, 1)
    out = torch.empty(plan.out_bound + 64, dtype=torch.uint8, device="cuda")
    prof = np.zeros(32, dtype=np.uint64)
    L.zgpu_prof_read(prof.ctypes.data, 1)
    plan.run(data.data_ptr(), out.data_ptr(), out.numel(), None)
    torch.cuda.synchronize()
    L.zgpu_prof_read(prof.ctypes.data, 1)
    t0 = time.perf_counter()
    plan.run(data.data_ptr(), out.data_ptr(), out.numel(), None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L.zgpu_prof_read(prof.ctypes.data, 1)
    off, _, _ = plan.fetch()
    nbytes = nu * ub
    per = prof[:16].astype(np.float64) / nbytes
    eper = prof[16:].astype(np.float64) / nu  # emit: cycles per unit
    sm, em = plan.phase_ms()
    print("%-8s %7.2f GB/s (search %.2f ms, emit %.2f ms) ratio %.4f | search cyc/B: %s | total %.1f" % (
        name, nbytes / dt / 1e9, sm, em, int(off[-1]) / nbytes,
        " ".join("%s=%.1f" % (p, v) for p, v in zip(PHASES, per) if p != "-"), per.sum()), flush=True)
    print("         emit cycles/unit: %s | total %.0f" % (" ".join("%s=%.0f" % (p, v) for p, v in zip(EPHASES, eper)), eper.sum()), flush=True)


def main():
    nu, ub = int(os.environ.get("NU", 2048)), int(os.environ.get("UB", 65536))
    if "LONG" in os.environ:  # "deep:fan"
        z._native.set_long_match(*[int(x) for x in os.environ["LONG"].split(":")])
    if "CHAIN" in os.environ:  # default: the library's own search depth
        z.set_search(int(os.environ["CHAIN"]), int(os.environ.get("NICE", "32")))
    dev = torch.device("cuda")
    full = bench.gen_units(int(os.environ.get("GEN", 20 * 512)), ub, 7, dev).view(-1, ub)
    kinds = {"text": (0, 10), "binary": (10, 15), "random": (15, 18), "lowent": (18, 20)}
    idx = torch.arange(full.shape[0], device=dev) % 20
    only = os.environ.get("KINDS")  # e.g. "binary,mixed"
    for name, (a, b) in kinds.items():
        if only and name not in only.split(","):
            continue
        rows = full[(idx >= a) & (idx < b)][:nu].contiguous()
        n = rows.shape[0]
        run(name, rows.view(-1), n, ub, 2, 0)
    if only and "mixed" not in only.split(","):
        return
    mixed = full[:nu].contiguous()
    run("mixed", mixed.view(-1), mixed.shape[0], ub, 2, 0)
    run("mixed-fx", mixed.view(-1), mixed.shape[0], ub, 1, 0)
    run("mixed-lz", mixed.view(-1), mixed.shape[0], ub, 2, 16)


if __name__ == "__main__":
    main()
