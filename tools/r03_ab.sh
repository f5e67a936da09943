set -e
mkdir -p gpurun_out
O=gpurun_out/r03_ab.txt
rm -f $O
REPS=2 ARGS="--steps 3" bash tools/ab_libs.sh libzgpu.so ../tools/libzgpu_nofan.bin >> $O 2>&1
