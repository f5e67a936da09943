# SQ LDS counters of the search kernel for two builds of the library (8192 units of the bench mix, one launch)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_ab
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib)
  ZGPU_LIB=$ROOT/$lib timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES -d "$OUT/$name" -o pmc --output-format csv -- python3 "$ROOT/bench.py" --no-cpu --no-extras --steps 1 --warmup 0 --units 8192 > "$OUT/$name.log" 2>&1 || exit 1
  f=$(find "$OUT/$name" -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$name" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Kernel_Name"].startswith("zgpu_deflate_kernel"):
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
print(sys.argv[2], " ".join("%s=%.4g" % kv for kv in sorted(agg.items())))
PY
done
