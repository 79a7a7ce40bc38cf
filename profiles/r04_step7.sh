#!/bin/bash
O=gpurun_out/r04_step7; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small_encode.py tests/test_gpu_file_io.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -12 $O/pytest.txt
grep -q "rc=0" $O/pytest.txt || exit 1
R=rev_5e3a44d
profiles/ab.sh r04s7 "$R 72 72 262144 mixed slots 10" "base 72 72 262144 mixed slots 10" "$R 72 72 262144 noise8 slots 10" "base 72 72 262144 noise8 slots 10" "$R 96 96 131072 mixed slots 10" "base 96 96 131072 mixed slots 10" "$R 128 128 65536 mixed slots 10" "base 128 128 65536 mixed slots 10" "$R 160 120 65536 mixed slots 10" "base 160 120 65536 mixed slots 10" "$R 160 120 65536 noise8 slots 10" "base 160 120 65536 noise8 slots 10" "$R 176 144 65536 mixed slots 10" "base 176 144 65536 mixed slots 10" "$R 200 200 32768 mixed slots 10" "base 200 200 32768 mixed slots 10" "$R 256 256 32768 mixed slots 10" "base 256 256 32768 mixed slots 10" "$R 320 240 16384 mixed slots 10" "base 320 240 16384 mixed slots 10" > $O/ab.txt 2>&1
cat $O/ab.txt
