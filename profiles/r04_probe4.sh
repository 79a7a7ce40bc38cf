#!/bin/bash
O=gpurun_out/r04_probe4; mkdir -p $O; : > $O/diag.txt
for kv in 0 128; do
  echo "DBDE_HIP_EXPERIMENT=$kv" >> $O/diag.txt
  DBDE_HIP_EXPERIMENT=$kv ABBENCH_DIAG=1 timeout -k 10 120 profiles/abbench profiles/variants/diag/libdbde_hip.so 1920 1080 512 mixed slots 1 diag 2>&1 | grep -E "trace" | cut -c1-600 >> $O/diag.txt
  DBDE_HIP_EXPERIMENT=$kv timeout -k 10 120 profiles/abbench profiles/variants/base/libdbde_hip.so 1920 1080 512 mixed slots 20 base 2>&1 | cut -c1-300 >> $O/diag.txt
  DBDE_HIP_EXPERIMENT=$kv timeout -k 10 120 profiles/abbench profiles/variants/base/libdbde_hip.so 2048 2048 1000 mixed concat 20 base 2>&1 | cut -c1-300 >> $O/diag.txt
done
cat $O/diag.txt
