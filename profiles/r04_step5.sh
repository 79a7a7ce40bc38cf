#!/bin/bash
O=gpurun_out/r04_step5; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_scatter.py tests/test_gpu_config5.py tests/test_gpu_streaming.py tests/test_c_client.py "tests/test_gpu_parity.py::test_config3_at_its_stated_size" "tests/test_gpu_parity.py::test_baseline_configs_full_size" -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -15 $O/pytest.txt
