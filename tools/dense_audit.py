#!/usr/bin/env python3
"""How often the dense-class pass takes units of real files, what it buys and what it costs: per set of
files of the image (8 MiB each, 64 KiB units), the units whose output differs with the pass on and off, the
summed sizes, and the time of the one batch call either way."""
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import zlib_es_amd as z  # noqa: E402

SETS = [("python sources", "/usr/lib/python3.10/**/*.py"), ("C headers", "/usr/include/**/*.h"),
        ("shared libraries", "/usr/lib/x86_64-linux-gnu/*.so*"), ("executables", "/usr/bin/*"),
        ("json", "/usr/local/lib/python3.10/dist-packages/**/*.json"), ("fonts", "/usr/share/fonts/**/*.ttf"),
        ("bytecode", "/usr/lib/python3.10/__pycache__/*.pyc"), ("rocm libraries", "/opt/rocm/lib/libhsa*.so*"),
        ("perl modules", "/usr/share/perl/**/*.pm"), ("numpy extension modules", "/usr/local/lib/python3.10/dist-packages/numpy/**/*.so")]
LIMIT = int(os.environ.get("LIMIT", 8 << 20))


def corpus(pattern):
    buf = bytearray()
    for f in sorted(glob.glob(pattern, recursive=True)):
        if os.path.isfile(f) and not os.path.islink(f):
            try:
                buf += open(f, "rb").read()
            except OSError:
                continue
        if len(buf) >= LIMIT:
            break
    return np.frombuffer(bytes(buf[:LIMIT]), dtype=np.uint8)


for name, pattern in SETS:
    d = corpus(pattern)
    units = [d[i:i + 65536] for i in range(0, d.size - 65535, 65536)]
    if not units:
        continue
    row = []
    for ct in (1, 2):
        res = {}
        for on in (1, 0):
            z._native.set_dense_search(z._native.DEFAULT_DENSE_CHAIN if on else 0)
            z.deflate_batch(units, ct, 0)
            t0 = time.perf_counter()
            outs, _, _ = z.deflate_batch(units, ct, 0)
            res[on] = ([len(o) for o in outs], (time.perf_counter() - t0) * 1e3)
        z._native.set_dense_search()
        taken = sum(a != b for a, b in zip(res[1][0], res[0][0]))
        row.append("ctype %d: %d of %d units taken, size %.4f of the size without the pass, %.1f ms against %.1f ms" % (
            ct, taken, len(units), sum(res[1][0]) / sum(res[0][0]), res[1][1], res[0][1]))
    print("%-24s %s | %s" % (name, row[0], row[1]), flush=True)
