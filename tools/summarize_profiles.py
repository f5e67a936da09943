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
    for sub, name in (("stats", "bench_kernel_stats"), ("ckstats", "checksum_kernel_stats"), ("zipstats", "zip_kernel_stats"),
                      ("gzipstats", "gzip_kernel_stats"), ("fixedstats", "fixed_kernel_stats")):
        p = find(sub, "kernel_stats.csv")
        if p:
            rows = list(csv.reader(open(p)))
            with open(os.path.join(DST, "%s_%s.csv" % (tag, name)), "w", newline="") as f:
                csv.writer(f).writerows(rows)
    # counters, per launch
    out = {}
    for sub in ("fetch", "write", "sq1", "sq2", "sq2_chain1", "sq2_deep0", "ckfetch"):
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
        # every kernel of a zgpu_plan_run (the bench's own verifier is not one)
        f_kb = sum(v["FETCH_SIZE"] for k, v in out["fetch"].items() if k != "zgpu_verify_kernel")
        w_kb = sum(v["WRITE_SIZE"] for k, v in out["write"].items() if k != "zgpu_verify_kernel")
        traffic = {
            "workload": {"units": 65536, "unit_bytes": 65536, "ctype": 2, "lazy": 0},
            "bytes_per_launch": int((2 * f_kb + w_kb) * 1024),
            "fetch_size_kb_raw": f_kb, "write_size_kb_raw": w_kb,
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), sum over the kernels of one "
                    "zgpu_plan_run (search and emit; no gathering pass any more), FETCH_SIZE doubled (gfx950 counts "
                    "128-B read requests at 64 B); the token stream between the two kernels is part of it",
            "source": "%s_pmc.json" % tag,
        }
        # LDS side of the search kernel (sq1 / sq2 passes, the same 65536-unit batch): share of the
        # kernel's cycles in which a CU's LDS pipe was busy, and in which it replayed bank conflicts.
        # SQ_BUSY_CYCLES is summed over the 32 shader engines; 256 CUs.
        try:
            q1, q2 = out["sq1"]["zgpu_deflate_kernel"], out["sq2"]["zgpu_deflate_kernel"]
            cu_cycles = q1["SQ_BUSY_CYCLES"] / 32.0 * 256.0
            nbytes = 65536 * 65536
            traffic["lds"] = {
                "kernel": "zgpu_deflate_kernel",
                "lds_pipe_busy_frac": round(q2["SQ_LDS_IDX_ACTIVE"] / cu_cycles, 4),
                "lds_bank_conflict_frac": round(q2["SQ_LDS_BANK_CONFLICT"] / cu_cycles, 4),
                "lds_wave_insts_per_input_byte": round(q1["SQ_INSTS_LDS"] / nbytes, 3),
                "valu_wave_insts_per_input_byte": round(q1["SQ_INSTS_VALU"] / nbytes, 3),
                "valu_issue_frac_of_peak": round(2.0 * q1["SQ_INSTS_VALU"] / q1["SQ_WAVE_CYCLES"], 4),
                "waves_in_valu_per_simd": round(4.0 * q1["SQ_ACTIVE_INST_VALU"] / q1["SQ_WAVE_CYCLES"], 3),
                "salu_wave_insts_per_input_byte": round(q1["SQ_INSTS_SALU"] / nbytes, 3),
                "wave_cycles_active_frac": round(q2["SQ_ACTIVE_INST_ANY"] / q1["SQ_WAVE_CYCLES"], 4),
                "wave_cycles_waitcnt_frac": round(q2["SQ_WAIT_ANY"] / q1["SQ_WAVE_CYCLES"], 4),
                "wave_cycles_wait_issue_frac": round(q2["SQ_WAIT_INST_ANY"] / q1["SQ_WAVE_CYCLES"], 4),
                "note": "SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT over CU-cycles (SQ_BUSY_CYCLES / 32 SEs x 256 CUs); "
                        "VALU: a SIMD-32 issues a wave64 instruction over 2 cycles (one wave alone sustains one per 4); "
                        "SQ_WAVE_CYCLES counts 4-cycle units summed over the 4 waves a SIMD holds, so it equals the "
                        "SIMD-cycles and the share of the SIMD's peak issue rate is 2 x SQ_INSTS_VALU / SQ_WAVE_CYCLES; "
                        "the wave_cycles_* shares split the waves' own time (SQ_ACTIVE_INST_ANY, SQ_WAIT_ANY, SQ_WAIT_INST_ANY)",
            }
            # LDS bank-conflict cycles with parts of the search switched off: what the chain walk and the deep walk add
            conf = {"default": q2["SQ_LDS_BANK_CONFLICT"]}
            for sub, name in (("sq2_chain1", "max_chain_1"), ("sq2_deep0", "deep_chain_0")):
                if sub in out and "zgpu_deflate_kernel" in out[sub]:
                    conf[name] = out[sub]["zgpu_deflate_kernel"]["SQ_LDS_BANK_CONFLICT"]
            traffic["lds"]["bank_conflict_cycles_by_setting"] = conf
        except KeyError:
            pass
        with open(os.path.join(DST, "traffic.json"), "w") as f:
            json.dump(traffic, f, indent=1)
    except KeyError:
        pass
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
