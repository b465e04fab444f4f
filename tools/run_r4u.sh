#!/bin/bash
# limiter kernel timing at 1 M and 10.1 M tets (kernel stats of bench.py), after the parity tests that cover them
o=gpurun_out/${1:-r4ad}; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_long_sedov.py tests/test_gpu_partition.py tests/test_gpu_edge_cases.py tests/test_gpu_multirank.py -m gpu -q -x > $o/pytest.log 2>&1; echo "rc $?" >> $o/pytest.log; tail -3 $o/pytest.log
grep -q "rc 0" $o/pytest.log || exit 1
bash tools/prof_stats.sh $o/nx55 > $o/nx55.log 2>&1; grep "superbee\|ms_per_step" $o/nx55.log | cut -c1-50,100-200
bash tools/prof_stats.sh $o/nx119 --nx 119 > $o/nx119.log 2>&1; grep "superbee\|ms_per_step" $o/nx119.log | cut -c1-50,100-200
