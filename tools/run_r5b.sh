#!/bin/bash
o=gpurun_out/${1:-r5b}; mkdir -p $o
timeout -k 10 1000 python -m pytest tests/test_gpu_two_layers.py tests/test_gpu_devmesh.py::test_overlapping_side_sets "tests/test_gpu_multirank.py::test_step_comm_as_a_hipgraph_and_with_two_ghost_layers" -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $o/pytest.log; tail -30 $o/pytest.log
