set -e
mkdir -p gpurun_out
O=gpurun_out/r03_ab.txt
rm -f $O
for w in zip gzip; do
echo "== $w default, then deep 0" >> $O
REPS=1 ARGS="--steps 3 --workload $w" bash tools/ab_libs.sh libzgpu.so >> $O 2>&1
REPS=1 ARGS="--steps 3 --workload $w --deep 0" bash tools/ab_libs.sh libzgpu.so >> $O 2>&1
done
