#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace --memory-copy-trace run of tools/host_path_trace.py: per kind of
event of the last call, count, summed and merged (union) time; usage: trace_summary.py <dir with the csv files>"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
k = list(csv.DictReader(open(glob.glob(d + "/*_kernel_trace.csv")[0])))
m = list(csv.DictReader(open(glob.glob(d + "/*_memory_copy_trace.csv")[0])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:28]) for r in k]
ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"]) for r in m]
ev.sort()
starts, prev_end = [ev[0][0]], ev[0][1]
for e in ev[1:]:
    if e[0] - prev_end > 3_000_000:
        starts.append(e[0])
    prev_end = max(prev_end, e[1])
t0 = starts[-1]
last = [e for e in ev if e[0] >= t0]


def union(evs):
    tot, cs, ce = 0, None, None
    for s, e in sorted(evs):
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return (tot + (ce - cs if cs is not None else 0)) / 1e6


by = collections.defaultdict(list)
for e in last:
    by[e[2]].append((e[0], e[1]))
for n, v in by.items():
    print("%-30s n=%4d sum %7.2f ms union %7.2f ms" % (n, len(v), sum(b - a for a, b in v) / 1e6, union(v)))
z = [(e[0], e[1]) for e in last if e[2].startswith("zgpu")]
print("zgpu kernels union %.2f ms; all events union %.2f ms; call %.2f ms" % (
    union(z), union([(e[0], e[1]) for e in last]), (max(e[1] for e in last) - t0) / 1e6))
if len(sys.argv) > 2:
    for e in last[: int(sys.argv[2])]:
        print("%8.3f %8.3f %s" % ((e[0] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[2]))
