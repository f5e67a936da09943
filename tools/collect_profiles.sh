#!/bin/bash
# Runs on the GPU box from the repository root: rocprofv3 kernel-trace stats of the
# default bench, then HBM-side and SQ counter passes (one counter group per run).
# Results land in gpurun_out/prof/; tools/summarize_profiles.py turns them into profiles/.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, timeout, rocprof args..., -- cmd
    local name=$1 to=$2; shift 2
    timeout -s KILL "$to" rocprofv3 "$@" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc"
    return $rc
}
run stats 300 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- python3 "$ROOT/bench.py" --no-cpu --no-extras || exit 1
run fetch 200 --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch --output-format csv -- python3 "$ROOT/bench.py" --no-cpu --no-extras --steps 1 --warmup 0 || exit 1
run write 200 --pmc WRITE_SIZE -d "$OUT/write" -o write --output-format csv -- python3 "$ROOT/bench.py" --no-cpu --no-extras --steps 1 --warmup 0 || exit 1
run sq1 200 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM -d "$OUT/sq1" -o sq1 --output-format csv -- python3 "$ROOT/bench.py" --no-cpu --no-extras --steps 1 --warmup 0 || exit 1
run sq2 200 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL -d "$OUT/sq2" -o sq2 --output-format csv -- python3 "$ROOT/bench.py" --no-cpu --no-extras --steps 1 --warmup 0 || exit 1
run ckstats 200 --kernel-trace --stats -d "$OUT/ckstats" -o ckstats --output-format csv -- python3 "$ROOT/bench.py" --workload checksum --no-cpu --steps 5 --warmup 2 || exit 1
run ckfetch 200 --pmc FETCH_SIZE -d "$OUT/ckfetch" -o ckfetch --output-format csv -- python3 "$ROOT/bench.py" --workload checksum --no-cpu --steps 1 --warmup 0 || exit 1
# where the LDS bank conflicts come from: the same counters with the chain walk cut to one candidate, and without the deep walk
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL"
run sq2_chain1 200 --pmc $SQ2 -d "$OUT/sq2_chain1" -o sq2 --output-format csv -- python3 "$ROOT/bench.py" --no-cpu --no-extras --steps 1 --warmup 0 --chain 1 || exit 1
run sq2_deep0 200 --pmc $SQ2 -d "$OUT/sq2_deep0" -o sq2 --output-format csv -- python3 "$ROOT/bench.py" --no-cpu --no-extras --steps 1 --warmup 0 --deep 0 || exit 1
run zipstats 300 --kernel-trace --stats -d "$OUT/zipstats" -o zipstats --output-format csv -- python3 "$ROOT/bench.py" --workload zip --no-cpu --no-extras || exit 1
run fixedstats 300 --kernel-trace --stats -d "$OUT/fixedstats" -o fixedstats --output-format csv -- python3 "$ROOT/bench.py" --ctype 1 --no-cpu --no-extras || exit 1
run gzipstats 300 --kernel-trace --stats -d "$OUT/gzipstats" -o gzipstats --output-format csv -- python3 "$ROOT/bench.py" --workload gzip --no-cpu --no-extras || exit 1
# keep what tools/summarize_profiles.py reads: the per-kernel statistics and the counters of this library's kernels
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*.db" -delete
for f in $(find "$OUT" -name "*counter_collection.csv"); do
    (head -1 "$f"; grep zgpu_ "$f") > "$f.tmp" && mv "$f.tmp" "$f"
done
du -sh "$OUT"
echo done
