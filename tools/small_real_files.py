#!/usr/bin/env python3
"""Small files: the first n bytes (300 .. 16000) of sixteen Python sources, JSON files, C headers and shared libraries
of the image, and synthetic text, one zgpu_deflate_batch call per set: sum of sizes against the reference's algorithm
(C restatement), the worst file, and how many files are above 1.03 x, FIXED and DYNAMIC.  ZGPU_KEY3_MAX=0 shows the
sizes with the chains of small FIXED units keyed on four bytes."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, glob
import _oracle as zo
import zlib_es_amd as z
def real(pattern, n, count):
    out = []
    for f in sorted(glob.glob(pattern, recursive=True)):
        if os.path.isfile(f) and os.path.getsize(f) >= n:
            out.append(np.fromfile(f, dtype=np.uint8, count=n))
            if len(out) == count: break
    return out
sets = {}
for n in (300, 1000, 3000, 8000, 16000):
    sets["text %d" % n] = [zo.gen({"gen": "text", "n": n, "seed": i}) for i in range(16)]
    sets["py %d" % n] = real("/usr/lib/python3.10/*.py", n, 16)
    sets["json %d" % n] = real("/usr/local/lib/python3.10/dist-packages/**/*.json", n, 16)
    sets["so %d" % n] = [d[4096:4096 + n] for d in real("/usr/lib/x86_64-linux-gnu/*.so*", 4096 + n, 16)]
    sets["hdr %d" % n] = real("/usr/include/*.h", n, 16)
for name, ds in sets.items():
    ds = [np.ascontiguousarray(d) for d in ds]
    if not ds: continue
    row = []
    for ct in (1, 2):
        outs, _, _ = z.deflate_batch(ds, ct, 0)
        rs = [len(zo.rawdeflate(d, ct)) for d in ds]
        ratios = [len(o) / r for o, r in zip(outs, rs)]
        row.append((sum(len(o) for o in outs) / sum(rs), max(ratios), sum(1 for x in ratios if x > 1.03)))
    print("%-10s FIXED sum %.3f worst %.3f above %d | DYNAMIC sum %.3f worst %.3f above %d  (%d files)" % (name, *row[0], *row[1], len(ds)), flush=True)
