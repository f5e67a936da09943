set -e
mkdir -p gpurun_out
rm -f gpurun_out/r03_exp1.txt
for L in 64:1; do
  echo "== LONG=$L" >> gpurun_out/r03_exp1.txt
  LONG=$L KINDS=binary,text timeout -k 10 120 python tools/phase_profile.py >> gpurun_out/r03_exp1.txt 2>&1
done
SETTINGS="0:1 64:1 96:1" timeout -k 10 200 python tools/long_match_sweep.py >> gpurun_out/r03_exp1.txt 2>&1
