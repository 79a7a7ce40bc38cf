#!/bin/bash
O=gpurun_out/r04_step3; mkdir -p $O
R=rev_5e3a44d
profiles/ab.sh r04s3 "$R 2048 2048 1000 mixed concat 20" "base 2048 2048 1000 mixed concat 20" "$R 1921 1081 2048 mixed slots 20" "base 1921 1081 2048 mixed slots 20" "noraw4 1921 1081 2048 mixed slots 20" "$R 1921 1081 2048 noise8 slots 20" "base 1921 1081 2048 noise8 slots 20" "noraw4 1921 1081 2048 noise8 slots 20" "$R 1920 1080 512 mixed slots 20" "base 1920 1080 512 mixed slots 20" "$R 1920 1080 2048 mixed slots 20" "base 1920 1080 2048 mixed slots 20" "$R 4096 3072 1024 mixed slots 10" "base 4096 3072 1024 mixed slots 10" "$R 4096 3072 1024 noise8 slots 10" "base 4096 3072 1024 noise8 slots 10" "$R 1001 1001 4096 mixed slots 10" "base 1001 1001 4096 mixed slots 10" "noraw4 1001 1001 4096 mixed slots 10" > $O/ab.txt 2>&1
cat $O/ab.txt
