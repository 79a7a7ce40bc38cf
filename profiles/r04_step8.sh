#!/bin/bash
O=gpurun_out/r04_step8; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_odd_widths.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
grep -q "rc=0" $O/pytest.txt || exit 1
R=rev_5e3a44d
ABBENCH_ONLY=dec profiles/ab.sh r04s8 "$R 1921 1081 2048 mixed slots 20" "base 1921 1081 2048 mixed slots 20" "$R 1921 1081 2048 noise8 slots 20" "base 1921 1081 2048 noise8 slots 20" "$R 1001 1001 4096 mixed slots 10" "base 1001 1001 4096 mixed slots 10" "$R 1366 768 4096 mixed slots 10" "base 1366 768 4096 mixed slots 10" "$R 1921 1081 2048 mixed slots 20" "base 1921 1081 2048 mixed slots 20" > $O/ab.txt 2>&1
cat $O/ab.txt
