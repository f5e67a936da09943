#!/usr/bin/env python3
"""Runs on the GPU box: compresses a set of named inputs through the product path (Python
mirror of the reference classes over the C ABI) and writes the streams with the SHA-256 of
their inputs to gpurun_out/gpu_streams.json.  Copied to tests/golden/gpu_streams.json, the
file lets the build container -- which holds the reference but no GPU -- run every stream
through the reference's own RawInflate / Inflate / Gunzip / Unzip
(tests/golden/check_gpu_streams.mjs)."""
import base64
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as zo  # noqa: E402
import zlib_es_amd as z  # noqa: E402

SPECS = [
    {"gen": "bytes", "hex": ""}, {"gen": "bytes", "hex": "41"}, {"gen": "bytes", "hex": "414141"},
    {"gen": "bytes", "hex": bytes(range(1, 24)).hex()}, {"gen": "bytes", "hex": b"abcabcabcabcXabcabc".hex()},
    {"gen": "seq", "n": 76543}, {"gen": "rnd", "n": 20000, "seed": 1},
    {"gen": "rndseq", "n": 76543, "seed": 1346432776267}, {"gen": "rndseq", "n": 65536, "seed": 1},
    {"gen": "text", "n": 65536, "seed": 42}, {"gen": "text", "n": 262144, "seed": 42},
    {"gen": "runs", "n": 65536, "seed": 3}, {"gen": "small-alpha", "n": 20000, "seed": 5, "k": 3},
    {"gen": "text", "n": 400000, "seed": 9},   # cut into pieces inside one call
]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def b64(b):
    return base64.b64encode(bytes(b)).decode()


def main():
    out = {"made_by": "tools/make_gpu_streams.py on an MI355X through libzgpu.so", "streams": []}
    add = out["streams"].append
    for spec in SPECS:
        d = zo.gen(spec)
        big = spec.get("n", 0) > 100000
        for ct in ((2,) if big else (1, 2)):
            for lazy in ((0, 8) if spec.get("n", 0) == 65536 else (0,)):
                raw = z.RawDeflate(d, {"compressionType": ct, "lazy": lazy}).compress()
                add({"kind": "raw", "spec": spec, "ctype": ct, "lazy": lazy, "n": int(d.size), "sha256": sha(d), "b64": b64(raw)})
        if big:
            continue
        add({"kind": "zlib", "spec": spec, "n": int(d.size), "sha256": sha(d), "b64": b64(z.Deflate(d).compress())})
        # the reference writes the comment only under flags.comment (src/gzip.js:95) while FCOMMENT
        # follows flags.fcomment (:72): both are set, or its own Gunzip cannot read the member
        add({"kind": "gzip", "spec": spec, "n": int(d.size), "sha256": sha(d),
             "b64": b64(z.Gzip(d, {"filename": "f.bin", "comment": "c", "flags": {"fname": True, "fcomment": True, "comment": True, "fhcrc": True}}).compress())})
    # stored blocks
    d = zo.gen({"gen": "text", "n": 70000, "seed": 3})
    add({"kind": "raw", "spec": {"gen": "text", "n": 70000, "seed": 3}, "ctype": 0, "lazy": 0, "n": int(d.size),
         "sha256": sha(d), "b64": b64(z.RawDeflate(d, {"compressionType": 0}).compress())})
    # one gzip member made of primed 64 KiB chunks, and of independent ones
    d = zo.gen({"gen": "text", "n": 300000, "seed": 11})
    for ind in (False, True):
        add({"kind": "gzip", "spec": {"gen": "text", "n": 300000, "seed": 11}, "chunked": 65536, "independent": ind,
             "n": int(d.size), "sha256": sha(d), "b64": b64(z.gzip_chunked(d, chunk_size=65536, mtime=0, independent=ind))})
    # a zip archive: stored, fixed and dynamic entries
    zp = z.Zip()
    entries = []
    for i, (spec, ct) in enumerate([({"gen": "text", "n": 5000, "seed": 1}, 2), ({"gen": "rndseq", "n": 30000, "seed": 2}, 1),
                                    ({"gen": "rnd", "n": 3000, "seed": 3}, 0), ({"gen": "text", "n": 90000, "seed": 4}, 2)]):
        d = zo.gen(spec)
        name = "entry%d.bin" % i
        opt = {"filename": list(name.encode())}
        if ct:
            opt["deflateOption"] = {"compressionType": ct}
        else:
            opt["compressionMethod"] = 0
        zp.addFile(d, opt)
        entries.append({"name": name, "n": int(d.size), "sha256": sha(d)})
    add({"kind": "zip", "entries": entries, "b64": b64(zp.compress())})
    path = os.path.join(ROOT, "gpurun_out", "gpu_streams.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out["streams"]), "streams")


main()
