#!/usr/bin/env python3
"""Size against the reference's algorithm on real files of the image (Python sources, licence
texts, shared libraries): 64 KiB units, DYNAMIC and FIXED, GPU path vs the C restatement.

usage: real_corpus_ratio.py [MAX_CHAIN ...]      (default: the library's own setting)"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as zo  # noqa: E402
import zlib_es_amd as z  # noqa: E402


def corpus(pattern, limit):
    buf = bytearray()
    for f in sorted(glob.glob(pattern, recursive=True)):
        if os.path.isfile(f) and not os.path.islink(f):
            try:
                buf += open(f, "rb").read()
            except OSError:
                continue
        if len(buf) >= limit:
            break
    return np.frombuffer(bytes(buf[:limit]), dtype=np.uint8)


SETS = [("python sources", "/usr/lib/python3.10/**/*.py", 8 << 20),
        ("licence texts", "/usr/share/common-licenses/*", 1 << 20),
        ("shared libraries", "/usr/lib/x86_64-linux-gnu/*.so*", 8 << 20),
        ("C headers", "/usr/include/**/*.h", 8 << 20)]
for name, pattern, limit in SETS:
    d = corpus(pattern, limit)
    if d.size < 65536:
        print(name, "not enough data")
        continue
    units = [d[i:i + 65536] for i in range(0, d.size - 65535, 65536)]
    step = max(1, len(units) // 48)
    sample = units[::step]
    for ct in (2, 1):
        refs = [len(zo.rawdeflate(u, ct, 0)) for u in sample]
        ref = sum(refs)
        res = []
        for chain in ([int(a) for a in sys.argv[1:]] or [z._native.DEFAULT_MAX_CHAIN]):
            z.set_search(chain)
            outs, _, _ = z.deflate_batch(sample, ct, 0)
            assert all(zo.inflate_raw(o, u.size) == u.tobytes() for o, u in zip(outs, sample))
            per = [len(o) / r for o, r in zip(outs, refs)]
            res.append("chain %d: %.4f (worst unit %.3f, %d above 1.03)" % (
                chain, sum(len(o) for o in outs) / ref, max(per), sum(x > 1.03 for x in per)))
        z.set_search()
        print("%-17s %4d units of 64 KiB, ctype %d, gpu/ref: %s (ref out/in %.3f)" % (
            name, len(sample), ct, ", ".join(res), ref / (len(sample) * 65536)), flush=True)
