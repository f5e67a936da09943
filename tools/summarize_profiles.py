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
    # HBM-side traffic of the default bench workload, per launch of the DEFLATE path
    # (search + emit kernels).  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE
    # reports half the bytes of wide (16 B per lane) streaming reads, so it is doubled.
    try:
        f_kb = sum(out["fetch"][k]["FETCH_SIZE"] for k in ("zgpu_deflate_kernel", "zgpu_emit_kernel"))
        w_kb = sum(out["write"][k]["WRITE_SIZE"] for k in ("zgpu_deflate_kernel", "zgpu_emit_kernel"))
        traffic = {
            "workload": {"units": 65536, "unit_bytes": 65536, "ctype": 2, "lazy": 0},
            "bytes_per_launch": int((2 * f_kb + w_kb) * 1024),
            "fetch_size_kb_raw": f_kb, "write_size_kb_raw": w_kb,
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), sum of zgpu_deflate_kernel and "
                    "zgpu_emit_kernel, FETCH_SIZE doubled (gfx950 counts 128-B read requests at 64 B); the token "
                    "stream between the two kernels is part of it",
            "source": "%s_pmc.json" % tag,
        }
        with open(os.path.join(DST, "traffic.json"), "w") as f:
            json.dump(traffic, f, indent=1)
    except KeyError:
        pass
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
