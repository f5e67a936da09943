#!/usr/bin/env python3
"""gpurun_out/prof/ (written by tools/collect_profiles.sh) -> profiles/<tag>_*.

usage: python tools/summarize_profiles.py r01
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")


def find(sub, suffix):
    hits = glob.glob(os.path.join(SRC, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def counters(sub):
    path = find(sub, "counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    if path:
        seen = set()
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                calls[k] += 1
    return agg, calls


def main():
    tag = sys.argv[1]
    os.makedirs(DST, exist_ok=True)
    # kernel stats
    for sub, name in (("stats", "bench_kernel_stats"), ("ckstats", "checksum_kernel_stats")):
        p = find(sub, "kernel_stats.csv")
        if p:
            rows = list(csv.reader(open(p)))
            with open(os.path.join(DST, "%s_%s.csv" % (tag, name)), "w", newline="") as f:
                csv.writer(f).writerows(rows)
    # counters, per launch
    out = {}
    for sub in ("fetch", "write", "sq1", "sq2", "ckfetch"):
        agg, calls = counters(sub)
        for k, v in agg.items():
            if not k.startswith("zgpu_"):
                continue
            for c, x in v.items():
                out.setdefault(sub, {}).setdefault(k, {})[c] = x / max(1, calls[k])
    with open(os.path.join(DST, "%s_pmc.json" % tag), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
