set -e
O=gpurun_out/r03_hostsweep.txt
rm -f $O
for sub in 67108864 134217728; do for sl in 3 4; do for cg in 1 2 3; do
echo "== SUB $sub SLOTS $sl GATE $cg" >> $O
ZGPU_COMPUTE_GATE=$cg ZGPU_SUB_BYTES=$sub ZGPU_SLOTS=$sl timeout -k 10 120 python tools/host_path_pinned.py 2>&1 | grep -v amdgpu | head -3 >> $O
done; done; done
