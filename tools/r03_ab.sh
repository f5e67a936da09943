set -e
mkdir -p gpurun_out
O=gpurun_out/r03_tail_first.txt
rm -f $O
echo "== bench, shipped walk against tail-first walk (same box, interleaved)" >> $O
REPS=2 ARGS="--steps 3" bash tools/ab_libs.sh libzgpu.so ../tools/libzgpu_tailfirst.bin >> $O 2>&1
echo "== phases on text units, shipped then tail-first" >> $O
KINDS=text timeout -k 10 120 python tools/phase_profile.py 2>&1 | grep -v "amdgpu\|emit cycles" >> $O
ZGPU_PROF_LIB=$PWD/tools/libzgpu_tailfirst_prof.bin KINDS=text timeout -k 10 120 python tools/phase_profile.py 2>&1 | grep -v "amdgpu\|emit cycles" >> $O
