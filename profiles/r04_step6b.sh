#!/bin/bash
O=gpurun_out/r04_step6; mkdir -p $O
timeout -k 10 900 python profiles/shim_threads.py > $O/shim_threads.jsonl 2> $O/shim_threads.err; echo "rc=$?"
cat $O/shim_threads.jsonl; tail -3 $O/shim_threads.err
