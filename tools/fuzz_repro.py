#!/usr/bin/env python3
"""Re-runs the configurations of tests/test_gpu_fuzz.py for one seed, several times, and
reports where a failing unit's stream first differs from its input (diagnostic).

usage: fuzz_repro.py SEED [REPS]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as zo  # noqa: E402
import test_gpu_fuzz as tf  # noqa: E402
import zlib_es_amd as z  # noqa: E402

seed = int(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = z._native
rng = np.random.default_rng(seed)
datas = [tf._unit(rng, i) for i in range(600)]
CONFIGS = [(2, 0, 8, 32, 0), (1, 0, 8, 258, 0), (2, 16, 3, 32, 0), (2, 0, 32, 258, 1), (1, 7, 1, 8, 1)]


def check(units, cfg, tag):
    ctype, lazy, chain, nice, chunked = cfg
    z.set_search(chain, nice)
    flags = (N.F_SYNC_ALIGN | N.F_FINAL_LAST) if chunked else N.F_FINAL_EACH
    outs, _, _ = z.deflate_batch(units, ctype, lazy, flags=flags)
    bad = []
    for i, (d, o) in enumerate(zip(units, outs)):
        o2 = o
        if chunked and i != len(units) - 1:
            o2 = o + b"\x01\x00\x00\xff\xff"
        try:
            got = zo.inflate_raw(o2, d.size)
            if got != d.tobytes():
                g = np.frombuffer(got, dtype=np.uint8)
                m = min(g.size, d.size)
                diff = np.nonzero(g[:m] != d[:m])[0]
                bad.append((i, d.size, "diffs %d first %s last %s got %d bytes" % (
                    diff.size, diff[:1].tolist(), diff[-1:].tolist(), g.size)))
        except Exception as e:  # noqa: BLE001
            bad.append((i, d.size, "inflate: %s" % e))
    if bad:
        print(tag, cfg, "bad:", bad[:8], flush=True)
    return bad


total = 0
for r in range(reps):
    for cfg in CONFIGS:
        total += len(check(datas, cfg, "rep %d" % r))
print("failures:", total)
