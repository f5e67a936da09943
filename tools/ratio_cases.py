#!/usr/bin/env python3
"""Per-case compressed size of the GPU path vs the reference sizes in
tests/golden/golden.json, for a few search depths."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as zo  # noqa: E402
import zlib_es_amd as z  # noqa: E402

golden = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
cases = [c for c in golden["cases"] if c["n"] >= 16384]
datas = [zo.gen(c["spec"]) for c in cases]
chains = [int(x) for x in os.environ.get("CHAINS", "8,16,32,128").split(",")]
LAZY = int(os.environ.get("LAZY", "0"))
NICE = int(os.environ.get("NICE", "32"))
for ct in (1, 2):
    print("ctype", ct)
    rows = {}
    for ch in chains:
        z.set_search(ch, NICE)
        outs, _, _ = z.deflate_batch(datas, ct, LAZY)
        rows[ch] = [len(o) for o in outs]
    for i, c in enumerate(cases):
        ref = c["raw"][str(ct)]["len"]
        print("  %-44s ref %8d " % (json.dumps(c["spec"])[:44], ref) +
              " ".join("c%d=%.3f" % (ch, rows[ch][i] / ref) for ch in chains))
    tot_ref = sum(c["raw"][str(ct)]["len"] for c in cases)
    print("  TOTAL " + " ".join("c%d=%.4f" % (ch, sum(rows[ch]) / tot_ref) for ch in chains))
