#!/bin/bash
O=gpurun_out/r04_full; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -6 $O/pytest.txt
