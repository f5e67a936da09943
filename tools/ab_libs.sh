#!/bin/bash
# Runs the default bench against several builds of the library on the same box, interleaved, and prints
# the search / emit kernel times of each: ZGPU_LIB selects the build.  usage: tools/ab_libs.sh lib1.so lib2.so ...
# (paths relative to zlib.es_amd/); REPS rounds (default 2); extra bench arguments in ARGS.
set -u
REPS=${REPS:-2}
for r in $(seq $REPS); do
  for lib in "$@"; do
    out=$(ZGPU_LIB=$PWD/zlib.es_amd/$lib timeout -k 10 300 python bench.py --no-cpu --no-extras ${ARGS:-} 2>/dev/null) || { echo "$lib failed"; exit 1; }
    python - "$lib" <<PY
import json,sys
r=json.loads('''$out''')
print(sys.argv[1], "value %.2f GB/s search %.2f ms emit %.2f ms out/in %.5f ok %s" % (r["value"], r["roofline"]["search_kernel_ms"], r["roofline"]["emit_kernel_ms"], r["ratio_out_over_in"], r.get("roundtrip_all_units")))
PY
  done
done
