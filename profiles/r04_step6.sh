#!/bin/bash
O=gpurun_out/r04_step6; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_c_client.py tests/test_gpu_reference_program.py tests/test_gpu_file_io.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -5 $O/pytest.txt
grep -q "rc=0" $O/pytest.txt || exit 1
timeout -k 10 900 python profiles/shim_threads.py > $O/shim_threads.jsonl 2> $O/shim_threads.err; echo "rc=$?"
cut -c1-125,190-330 $O/shim_threads.jsonl; tail -3 $O/shim_threads.err
