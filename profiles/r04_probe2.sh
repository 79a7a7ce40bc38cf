#!/bin/bash
# round 4: per-workgroup timeline of persistent-encoder launches (-DDBDE_DIAG build)
O=gpurun_out/r04_probe2; mkdir -p $O
for spec in "1920 1080 512 mixed slots" "2048 2048 1000 mixed concat" "1921 1081 2048 mixed slots" "1920 1080 128 mixed slots" "4096 3072 64 noise8 slots"; do
  ABBENCH_DIAG=1 timeout -k 10 120 profiles/abbench profiles/variants/diag/libdbde_hip.so $spec 1 diag >> $O/diag.txt 2>&1
done
cat $O/diag.txt
