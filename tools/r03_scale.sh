set -e
( while true; do date >> gpurun_out/heartbeat.txt; sleep 30; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 300 python bench.py --workload zip --no-extras > gpurun_out/r03_zip_full.json 2>/dev/null
timeout -k 10 300 python bench.py --workload gzip --no-extras > gpurun_out/r03_gzip_full.json 2>/dev/null
timeout -k 10 500 python bench.py --workload zip --units 1048576 --no-extras > gpurun_out/r03_scale_zip_1m_files.json 2>/dev/null
timeout -k 10 800 python bench.py --workload gzip --units 65536 --no-extras --no-cpu > gpurun_out/r03_scale_gzip_64gib.json 2>/dev/null
