for kb in 1 4 8 16 64 256 1024 4096; do
  u=$((1048576/kb)); timeout -k 10 120 python bench.py --units $u --unit-bytes $((kb*1024)) --no-cpu --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print($kb, round(d['value'],1))" || exit 1
done
