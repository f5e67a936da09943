#!/bin/bash
# 1 GiB of the bench mix per run at unit sizes 1 KiB .. 4 MiB (DYNAMIC): unit KiB, GB/s, search ms, emit ms,
# step ms, out/in, all units verified.  Runs on the GPU box from the repository root.
for kb in 1 2 4 8 16 64 256 1024 4096; do
  u=$((1048576/kb)); timeout -k 10 160 python bench.py --units $u --unit-bytes $((kb*1024)) --no-cpu --no-extras --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print($kb, round(d['value'],1), r['search_kernel_ms'], r['emit_kernel_ms'], d['ms_per_step'], d['ratio_out_over_in'], d.get('roundtrip_all_units'))" || exit 1
done
