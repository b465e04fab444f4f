#!/bin/bash
# parity tests that cover the limiter / step kernels, then kernel stats at 1 M and 10.1 M tets
o=gpurun_out/${1:-r4ad}; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_long_sedov.py tests/test_gpu_partition.py tests/test_gpu_edge_cases.py tests/test_gpu_multirank.py tests/test_gpu_amr.py tests/test_gpu_cpp_adapter.py -m gpu -q -x > $o/pytest.log 2>&1; echo "rc $?" >> $o/pytest.log; tail -3 $o/pytest.log
grep -q "rc 0" $o/pytest.log || exit 1
bash tools/prof_stats.sh $o/nx55 > $o/nx55.log 2>&1
bash tools/prof_stats.sh $o/nx119 --nx 119 > $o/nx119.log 2>&1
for v in nx55 nx119; do echo "== $v $(grep -o '"ms_per_step": [0-9.]*' $o/$v/bench.log | head -1)"; grep -h "superbee\|k_rhs_p1w" $o/$v/*/*kernel_stats.csv | awk -F'",' '{split($2,a,","); printf "   %-44s %10.1f us\n", substr($1,12,44), a[3]/1000}'; done
