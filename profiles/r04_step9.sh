#!/bin/bash
O=gpurun_out/r04_step9; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_odd_widths.py tests/test_gpu_parity.py tests/test_gpu_small_encode.py tests/test_gpu_fused_decode.py tests/test_gpu_u16.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
grep -q "rc=0" $O/pytest.txt || exit 1
R=rev_5e3a44d
profiles/ab.sh r04s9 "$R 2048 2048 1000 mixed concat 20" "base 2048 2048 1000 mixed concat 20" "$R 1921 1081 2048 mixed slots 20" "base 1921 1081 2048 mixed slots 20" "$R 1920 1080 512 mixed slots 20" "base 1920 1080 512 mixed slots 20" "$R 1920 1080 2048 mixed slots 20" "base 1920 1080 2048 mixed slots 20" "$R 4096 3072 1024 noise8 slots 10" "base 4096 3072 1024 noise8 slots 10" "$R 1921 1081 2048 noise8 slots 20" "base 1921 1081 2048 noise8 slots 20" > $O/ab.txt 2>&1
cat $O/ab.txt
for spec in "1920 1080 512 mixed slots" "2048 2048 1000 mixed concat"; do
  ABBENCH_DIAG=1 timeout -k 10 120 profiles/abbench profiles/variants/diag/libdbde_hip.so $spec 1 diag 2>&1 | grep -E "trace" | cut -c1-600 >> $O/diag.txt
done
DBDE_HIP_EXPERIMENT=1 timeout -k 10 120 profiles/abbench profiles/variants/base/libdbde_hip.so 1920 1080 512 mixed slots 5 tickets 2>&1 | cut -c1-250 >> $O/diag.txt
cat $O/diag.txt
for v in rev_5e3a44d base; do
DBDE_HIP_EXPERIMENT=1 timeout -k 10 120 profiles/abbench profiles/variants/$v/libdbde_hip.so 1920 1080 512 mixed slots 5 tickets_$v 2>&1 | cut -c1-250
done
