#!/bin/bash
# k1 (fewer VALU in multi_rounds) / k2 (+ uniform regions left unstructured) / k3 (+ immediate waits) against r4:
# uniform kernels through kbench (one process, interleaved), mixed launches through bench.py (one process per library)
cd "$(dirname "$0")/../.."
R=$(pwd)
L="$R/tools/libafsk_r4.so:$R/tools/libafsk_k1.so:$R/tools/libafsk_k2.so"
bash tools/r5_exp2.sh "$L" "1200 6000 12000 4000 3000 2000 300 2400 375 160 800 96" 65536 2>&1 | grep -v "two-pass\|rotating" | tee gpurun_out/r5_exp3_uniform.txt
bash tools/r5_exp2.sh "$L" "1200 6000 375" 4096 2>&1 | grep -v "two-pass\|rotating" | tee -a gpurun_out/r5_exp3_uniform.txt
for args in "--workload config3" "--workload custom --bauds 375,160,96,1200 --streams 65536" "--workload custom --bauds 375,160,96,1200 --streams 4096 --steps 200"; do
  echo "=== $args" | tee -a gpurun_out/r5_exp3_mixed.txt
  bash tools/lib_ab.sh "$args" $R/tools/libafsk_r4.so $R/tools/libafsk_k1.so $R/tools/libafsk_k2.so $R/tools/libafsk_k3.so 2>&1 | tee -a gpurun_out/r5_exp3_mixed.txt
done
