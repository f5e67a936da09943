"""The named inputs and settings behind tests/golden/gpu_streams.json, and the function that
compresses them through the product path on a GPU: the Python mirror of the reference classes
over the C ABI, and the JavaScript shim over the N-API addon under Node.  tools/make_gpu_streams.py
writes the fixture with it; tests/test_gpu_parity_golden.py regenerates the streams and compares
them with the fixture byte for byte."""
import base64
import hashlib
import json
import os
import shutil
import subprocess
import tempfile

import _oracle as zo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JS = os.path.join(ROOT, "zlib.es_amd", "js")

SPECS = [
    {"gen": "bytes", "hex": ""}, {"gen": "bytes", "hex": "41"}, {"gen": "bytes", "hex": "414141"},
    {"gen": "bytes", "hex": bytes(range(1, 24)).hex()}, {"gen": "bytes", "hex": b"abcabcabcabcXabcabc".hex()},
    {"gen": "seq", "n": 76543}, {"gen": "rnd", "n": 20000, "seed": 1},
    {"gen": "rndseq", "n": 76543, "seed": 1346432776267}, {"gen": "rndseq", "n": 65536, "seed": 1},
    {"gen": "text", "n": 65536, "seed": 42}, {"gen": "text", "n": 262144, "seed": 42},
    {"gen": "runs", "n": 65536, "seed": 3}, {"gen": "small-alpha", "n": 20000, "seed": 5, "k": 3},
    {"gen": "text", "n": 400000, "seed": 9},   # cut into pieces inside one call
]
# search settings other than the defaults: (name, max_chain, nice_length, deep_chain, probe_fans)
VARIANTS = [("thorough", 8, 64, 0, 1), ("long-match", 8, 32, 128, 3), ("no-deep-walk", 8, 32, 0, 1)]
VARIANT_SPECS = [{"gen": "rndseq", "n": 76543, "seed": 1346432776267}, {"gen": "text", "n": 65536, "seed": 42},
                 {"gen": "runs", "n": 65536, "seed": 3}]
JS_SPECS = [{"gen": "bytes", "hex": b"abcabcabcabcXabcabc".hex()}, {"gen": "text", "n": 65536, "seed": 42},
            {"gen": "rndseq", "n": 76543, "seed": 1346432776267}, {"gen": "text", "n": 400000, "seed": 9}]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def b64(b):
    return base64.b64encode(bytes(b)).decode()


def _js_streams(add):
    """Streams made by zlib_gpu.js (reference class API over the addon) under Node."""
    node = shutil.which("node")
    if node is None:
        raise RuntimeError("node is not installed: the JavaScript streams cannot be made")
    with tempfile.TemporaryDirectory() as tmp:
        jobs = []
        for i, spec in enumerate(JS_SPECS):
            d = zo.gen(spec)
            path = os.path.join(tmp, "in%d.bin" % i)
            with open(path, "wb") as f:
                f.write(bytes(d))
            for kind in ("raw", "zlib", "gzip"):
                jobs.append({"file": path, "kind": kind, "spec": spec, "n": int(d.size), "sha256": sha(d)})
        # one archive of the first two inputs under a password, through the shim's Zip
        pw = list(b"hogefugapiyo")
        jobs.append({"kind": "zip", "password": pw, "files": [os.path.join(tmp, "in0.bin"), os.path.join(tmp, "in1.bin")],
                     "names": ["js0.bin", "js1.bin"],
                     "entries": [{"name": "js%d.bin" % i, "n": int(zo.gen(JS_SPECS[i]).size), "sha256": sha(zo.gen(JS_SPECS[i]))}
                                 for i in range(2)]})
        with open(os.path.join(tmp, "jobs.json"), "w") as f:
            json.dump(jobs, f)
        r = subprocess.run([node, os.path.join(JS, "test", "make_streams.js"), os.path.join(tmp, "jobs.json")],
                           cwd=JS, capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            raise RuntimeError("make_streams.js failed: " + r.stdout[-2000:] + r.stderr[-2000:])
        outs = json.loads(r.stdout.strip().splitlines()[-1])
        for job, o in zip(jobs, outs):
            if job["kind"] == "zip":
                add({"kind": "zip", "via": "js", "password": job["password"], "entries": job["entries"], "b64": o})
                continue
            add({"kind": job["kind"], "via": "js", "spec": job["spec"], "ctype": 2, "lazy": 0, "n": job["n"],
                 "sha256": job["sha256"], "b64": o})


def make_streams():
    import zlib_es_amd as z
    out = {"made_by": "tests/_gpu_streams.py on an MI355X through libzgpu.so (Python mirror and zlib_gpu.js)", "streams": []}
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
    # the search settings a caller can change
    for name, chain, nice, deep, fan in VARIANTS:
        z.set_search(chain, nice)
        z._native.set_long_match(deep, fan)
        try:
            for spec in VARIANT_SPECS:
                d = zo.gen(spec)
                for ct in (1, 2):
                    raw = z.RawDeflate(d, {"compressionType": ct}).compress()
                    add({"kind": "raw", "variant": name, "spec": spec, "ctype": ct, "lazy": 0, "n": int(d.size), "sha256": sha(d), "b64": b64(raw)})
        finally:
            z.set_search()
            z._native.set_long_match()
    # a lazy threshold below the library's floor takes effect once the floor is lowered
    z._native.set_min_lazy(0)
    try:
        for spec in VARIANT_SPECS:
            d = zo.gen(spec)
            raw = z.RawDeflate(d, {"compressionType": 2, "lazy": 32}).compress()
            add({"kind": "raw", "variant": "min-lazy-0", "spec": spec, "ctype": 2, "lazy": 32, "n": int(d.size), "sha256": sha(d), "b64": b64(raw)})
    finally:
        z._native.set_min_lazy(258)
    # stored blocks
    d = zo.gen({"gen": "text", "n": 70000, "seed": 3})
    add({"kind": "raw", "spec": {"gen": "text", "n": 70000, "seed": 3}, "ctype": 0, "lazy": 0, "n": int(d.size),
         "sha256": sha(d), "b64": b64(z.RawDeflate(d, {"compressionType": 0}).compress())})
    # one gzip member / one zlib stream made of 64 KiB chunks, primed with the bytes before them or independent
    d = zo.gen({"gen": "text", "n": 300000, "seed": 11})
    for ind in (False, True):
        add({"kind": "gzip", "spec": {"gen": "text", "n": 300000, "seed": 11}, "chunked": 65536, "independent": ind,
             "n": int(d.size), "sha256": sha(d), "b64": b64(z.gzip_chunked(d, chunk_size=65536, mtime=0, independent=ind))})
        add({"kind": "zlib", "spec": {"gen": "text", "n": 300000, "seed": 11}, "chunked": 65536, "independent": ind,
             "n": int(d.size), "sha256": sha(d), "b64": b64(z.zlib_chunked(d, chunk_size=65536, independent=ind))})
    # a zip archive: stored, fixed and dynamic entries
    zp = z.Zip()
    entries = []
    for i, (spec, ct) in enumerate([({"gen": "text", "n": 5000, "seed": 1}, 2), ({"gen": "rndseq", "n": 30000, "seed": 2}, 1),
                                    ({"gen": "rnd", "n": 3000, "seed": 3}, 0), ({"gen": "text", "n": 90000, "seed": 4}, 2)]):
        d = zo.gen(spec)
        name = "entry%d.bin" % i
        opt = {"filename": list(name.encode()), "date": (2024, 1, 2, 3, 4, 6)}  # the default is the time of the call
        if ct:
            opt["deflateOption"] = {"compressionType": ct}
        else:
            opt["compressionMethod"] = 0
        zp.addFile(d, opt)
        entries.append({"name": name, "n": int(d.size), "sha256": sha(d)})
    add({"kind": "zip", "entries": entries, "b64": b64(zp.compress())})
    # an archive under a password (traditional PKZIP encryption on the host, src/zip.js:138-147; the twelve
    # header bytes per entry come from a pinned generator so that the archive can be regenerated)
    st = [20121006]

    def rnd():
        st[0] = st[0] * 48271 % 2147483647
        return st[0] / 2147483647
    pw = list(b"hogefugapiyo")
    zp = z.Zip()
    zp.random = rnd
    zp.setPassword(pw)
    entries = []
    for i, (spec, ct) in enumerate([({"gen": "text", "n": 20000, "seed": 21}, 2), ({"gen": "rndseq", "n": 9000, "seed": 22}, 1),
                                    ({"gen": "rnd", "n": 500, "seed": 23}, 0)]):
        d = zo.gen(spec)
        name = "secret%d.bin" % i
        opt = {"filename": list(name.encode()), "date": (2024, 1, 2, 3, 4, 6)}
        if ct:
            opt["deflateOption"] = {"compressionType": ct}
        else:
            opt["compressionMethod"] = 0
        zp.addFile(d, opt)
        entries.append({"name": name, "n": int(d.size), "sha256": sha(d)})
    add({"kind": "zip", "password": pw, "entries": entries, "b64": b64(zp.compress())})
    _js_streams(add)
    return out
