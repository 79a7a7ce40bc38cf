#!/bin/bash
O=gpurun_out/r04_probe3; mkdir -p $O
for kv in 0 1; do
  echo "HIP_FORCE_DEV_KERNARG=$kv" >> $O/diag.txt
  HIP_FORCE_DEV_KERNARG=$kv ABBENCH_DIAG=1 timeout -k 10 120 profiles/abbench profiles/variants/diag/libdbde_hip.so 1920 1080 512 mixed slots 1 diag 2>&1 | grep -E "trace|enc_ms" | cut -c1-600 >> $O/diag.txt
  HIP_FORCE_DEV_KERNARG=$kv timeout -k 10 120 profiles/abbench profiles/variants/base/libdbde_hip.so 1920 1080 512 mixed slots 20 base 2>&1 | cut -c1-300 >> $O/diag.txt
done
cat $O/diag.txt
