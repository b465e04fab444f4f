#!/bin/bash
# full GPU suite + self-halo bench (depth 1 / depth 2, plain / graph)
o=gpurun_out/${1:-r5c}; mkdir -p $o
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $o/pytest.log; tail -5 $o/pytest.log
