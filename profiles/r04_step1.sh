#!/bin/bash
# round 4: parity subset on the new encoder prologue / raw4 fetches, then A/B lines and the per-workgroup timeline
O=gpurun_out/r04_step1; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_odd_widths.py tests/test_gpu_parity.py tests/test_gpu_small_encode.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -5 $O/pytest.txt
grep -q "rc=0" $O/pytest.txt || exit 1
profiles/ab.sh r04s1 "base 2048 2048 1000 mixed concat 20" "base 1921 1081 2048 mixed slots 20" "noraw4 1921 1081 2048 mixed slots 20" "base 1921 1081 6144 mixed slots 10" "base 1921 1081 2048 noise8 slots 20" "noraw4 1921 1081 2048 noise8 slots 20" "base 1920 1080 2048 mixed slots 20" "base 1920 1080 512 mixed slots 20" "base 4096 3072 1024 mixed slots 10" "base 4096 3072 1024 noise8 slots 10" "base 1366 768 4096 mixed slots 10" "noraw4 1366 768 4096 mixed slots 10" "base 1001 1001 4096 mixed slots 10" "noraw4 1001 1001 4096 mixed slots 10" > $O/ab.txt 2>&1
cat $O/ab.txt
for spec in "1920 1080 512 mixed slots" "2048 2048 1000 mixed concat" "1921 1081 2048 mixed slots"; do
  ABBENCH_DIAG=1 timeout -k 10 120 profiles/abbench profiles/variants/diag/libdbde_hip.so $spec 1 diag 2>&1 | grep -E "trace|enc_ms" >> $O/diag.txt
done
cat $O/diag.txt
