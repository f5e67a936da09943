#!/usr/bin/env python3
"""The 64 KiB units of real files (tools/real_corpus_ratio.py's sets) whose size is more than 3 % above the
reference's: what the classifier's statistics say about them, and what other settings make of them."""
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


def stats(d, chunks=32):
    n = d.size
    stride = (n - 65) // (chunks - 1)
    idx = np.arange(chunks)[:, None] * stride + np.arange(64)[None, :]
    b0, b1 = d[idx], d[idx + 1]
    cnt = np.bincount(b0.reshape(-1), minlength=256).astype(np.int64)
    return (cnt * cnt).sum() / (64 * chunks) ** 2, (b0 == b1).mean(), (b0 == b1).all(axis=1).mean()


SETS = [("shared libraries", "/usr/lib/x86_64-linux-gnu/*.so*", int(os.environ.get("LIMIT", 16 << 20))),
        ("C headers", "/usr/include/**/*.h", 8 << 20)]
if "SETS" in os.environ:  # "name=glob,name=glob": other files of the image, 8 MiB of each
    SETS = [(kv.split("=")[0], kv.split("=")[1], int(os.environ.get("LIMIT", 8 << 20))) for kv in os.environ["SETS"].split(",")]
if "DENSE" in os.environ:  # chain budget of the dense-class pass
    z._native.set_dense_search(int(os.environ["DENSE"]))
BRIEF = "BRIEF" in os.environ  # counts and the worst units only, at the default settings
SETTINGS = [("default", None), ("chain 32", (32, 32)), ("thorough 8", (8, 258)), ("thorough 2048", (2048, 258)),
            ("exhaustive", (32768, 258))]
for name, pattern, limit in SETS:
    d = corpus(pattern, limit)
    units = [d[i:i + 65536] for i in range(0, d.size - 65535, 65536)]
    if not units:
        print(name, "not enough data")
        continue
    for ct in (1, 2):
        refs = [len(zo.rawdeflate(u, ct, 0)) for u in units]
        outs, _, _ = z.deflate_batch(units, ct, 0)
        bad = [i for i, (o, r) in enumerate(zip(outs, refs)) if len(o) > 1.03 * r]
        print("%s, ctype %d: %d units, %d above 1.03, sum %.4f" % (
            name, ct, len(units), len(bad), sum(len(o) for o in outs) / sum(refs)), flush=True)
        if not bad:
            continue
        if BRIEF:
            worst = sorted(bad, key=lambda i: -len(outs[i]) / refs[i])[:4]
            for i in worst:
                sq, eq, one = stats(units[i])
                print("  unit %4d ref out/in %.3f squares %.3f equal %.2f one-value %.2f | default %.3f | %s" % (
                    i, refs[i] / 65536, sq, eq, one, len(outs[i]) / refs[i], bytes(units[i][32768:32768 + 24]).hex()), flush=True)
            continue
        sub = [units[i] for i in bad]
        cols = {}
        for sname, st in SETTINGS:
            if st:
                z.set_search(*st)
            o2, _, _ = z.deflate_batch(sub, ct, 0)
            z.set_search()
            cols[sname] = [len(o) for o in o2]
        z._native.set_min_lazy(0)
        o2, _, _ = z.deflate_batch(sub, ct, 0)
        z._native.set_min_lazy(258)
        cols["greedy"] = [len(o) for o in o2]
        for j, i in enumerate(bad):
            sq, eq, one = stats(units[i])
            print("  unit %4d ref out/in %.3f squares %.3f equal %.2f one-value %.2f | %s" % (
                i, refs[i] / 65536, sq, eq, one,
                " ".join("%s %.3f" % (k, v[j] / refs[i]) for k, v in cols.items())), flush=True)
