#!/bin/bash
for v in rev_5e3a44d base; do
DBDE_HIP_EXPERIMENT=1 timeout -k 10 120 profiles/abbench profiles/variants/$v/libdbde_hip.so 1920 1080 512 mixed slots 5 tickets_$v 2>&1 | cut -c1-250
DBDE_HIP_EXPERIMENT=1 timeout -k 10 120 profiles/abbench profiles/variants/$v/libdbde_hip.so 2048 2048 1000 mixed concat 5 tickets_$v 2>&1 | cut -c1-250
done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ticket or fallback or persistent" 2>&1 | tail -3
