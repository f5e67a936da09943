#!/usr/bin/env python3
"""Token statistics (blocks, literals, matches) of the GPU stream vs the reference's
for a few inputs and parse settings -- a diagnostic for size differences."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as zo  # noqa: E402
import zlib_es_amd as z  # noqa: E402



def toks(stream, n):
    import ctypes
    import numpy as np
    zo.inflate_raw(stream, n)
    out = np.zeros(2, dtype=np.uint32)
    zo.lib().zo_inflate_last_tokens(ctypes.c_void_p(out.ctypes.data))
    return out.tolist()


specs = [{"gen": "rndseq", "n": 65536, "seed": 1}, {"gen": "runs", "n": 65536, "seed": 3}, {"gen": "seq", "n": 76543},
         {"gen": "small-alpha", "n": 70000, "seed": 6, "k": 3}, {"gen": "text", "n": 65536, "seed": 42}]
for spec in specs:
    d = zo.gen(spec)
    ref = zo.rawdeflate(d, 2)
    c = toks(ref, d.size)
    print(spec, "ref bytes", len(ref), "(literals, matches)", c)
    for nice in (32, 258):
        for chain in (8, 128):
            min_lazy = 258
            z.set_search(chain, nice)
            (o,), _, _ = z.deflate_batch([d], 2, 0)
            c = toks(o, d.size)
            print("   nice %3d chain %3d: bytes %6d (%.3f) (literals, matches) %s" % (nice, chain, len(o), len(o) / len(ref), c))
z._native.set_min_lazy(258)
z.set_search()
