set -e
mkdir -p gpurun_out
O=gpurun_out/r03_ab.txt
rm -f $O
REPS=2 ARGS="--steps 3" bash tools/ab_libs.sh ../tools/libzgpu_base.bin ../tools/libzgpu_dense.bin >> $O 2>&1
SETTINGS="96:0" timeout -k 10 300 python tools/long_match_sweep.py >> $O 2>&1
