#!/bin/bash
O=gpurun_out/r04_step10; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
grep -q "rc=0" $O/pytest.txt || exit 1
for v in rev_5e3a44d base; do
DBDE_HIP_EXPERIMENT=1 timeout -k 10 120 profiles/abbench profiles/variants/$v/libdbde_hip.so 1920 1080 512 mixed slots 5 tickets_$v 2>&1 | cut -c1-250
done
R=rev_5e3a44d
ABBENCH_ONLY=dec profiles/ab.sh r04s10 "$R 1921 1081 2048 mixed slots 20" "base 1921 1081 2048 mixed slots 20" "$R 1921 1081 2048 noise8 slots 20" "base 1921 1081 2048 noise8 slots 20" "$R 1001 1001 4096 mixed slots 10" "base 1001 1001 4096 mixed slots 10" "$R 1366 768 4096 mixed slots 10" "base 1366 768 4096 mixed slots 10" "$R 1921 1081 2048 mixed slots 20" "base 1921 1081 2048 mixed slots 20" 2>&1 | grep -v "^$"
