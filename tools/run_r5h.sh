#!/bin/bash
o=gpurun_out/${1:-r5h}; mkdir -p $o
timeout -k 10 1000 python -m pytest tests/test_gpu_amr.py tests/test_gpu_cpp_adapter.py tests/test_gpu_two_layers.py -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $o/pytest.log; tail -30 $o/pytest.log
