#!/usr/bin/env python3
"""Build container only: times the reference's RawDeflate under Node (tools/time_reference.mjs) and the C
restatement (oracle/) on the same 1 MiB of word-model text, one core each, and writes
profiles/<tag>_node_reference_timing.json.   usage: python tools/time_reference.py r02"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as zo  # noqa: E402


def best(fn, k=4):
    b, r = 1e30, None
    for _ in range(k):
        t = time.perf_counter()
        r = fn()
        b = min(b, time.perf_counter() - t)
    return b, r


def main():
    tag = sys.argv[1]
    spec = {"gen": "text", "n": 1 << 20, "seed": 42}
    d = zo.gen(spec)
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(bytes(d))
        path = f.name
    try:
        r = subprocess.run(["node", os.path.join(ROOT, "tools", "time_reference.mjs"), path], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        ref = json.loads(r.stdout.strip().splitlines()[-1])
    finally:
        os.unlink(path)
    port = {}
    for ct in (1, 2):
        tw, whole = best(lambda: len(zo.rawdeflate(d, ct, 0)))
        tu, units = best(lambda: sum(len(zo.rawdeflate(d[o:o + 65536], ct, 0)) for o in range(0, d.size, 65536)))
        port["ctype%d" % ct] = {"whole_s": round(tw, 4), "whole_MBps": round(d.size / tw / 1e6, 2), "whole_out": whole,
                                "units64k_s": round(tu, 4), "units64k_MBps": round(d.size / tu / 1e6, 2), "units64k_out": units}
    out = {"input": spec, "cores": 1, "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
           "reference_under_node": ref, "c_restatement": port,
           "same_output_sizes": all(ref[k]["whole_out"] == port[k]["whole_out"] and ref[k]["units64k_out"] == port[k]["units64k_out"] for k in port)}
    p = os.path.join(ROOT, "profiles", "%s_node_reference_timing.json" % tag)
    json.dump(out, open(p, "w"), indent=1)
    print(json.dumps(out))


main()
