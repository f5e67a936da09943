set -e
mkdir -p gpurun_out
O=gpurun_out/r03_ab.txt
rm -f $O
REPS=2 ARGS="--steps 3" bash tools/ab_libs.sh ../tools/libzgpu_p3l54.bin ../tools/libzgpu_p3l60.bin ../tools/libzgpu_p4l60.bin ../tools/libzgpu_p4l63.bin >> $O 2>&1
