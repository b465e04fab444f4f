#!/bin/bash
# round 4: limiter kernels at higher occupancy (means-only LDS, half-tile store staging) vs the previous form
o=gpurun_out/r4w; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_long_sedov.py tests/test_gpu_partition.py tests/test_gpu_edge_cases.py tests/test_gpu_multirank.py tests/test_gpu_config4.py -m gpu -q -x > $o/pytest.log 2>&1; echo "rc $?" >> $o/pytest.log; tail -3 $o/pytest.log
grep -q "rc 0" $o/pytest.log || exit 1
for nx in 55 119; do
  extra=""; [ $nx = 119 ] && extra="--nx 119"
  bash tools/prof_stats.sh $o/new$nx $extra > $o/new$nx.log 2>&1
  QDG_LIB=$PWD/quinoa_amd/lib/variants/sb_v1/libqdg.so bash tools/prof_stats.sh $o/old$nx $extra > $o/old$nx.log 2>&1
  for v in new old; do echo "== $v nx $nx"; grep "superbee\|k_rhs_p1w" $o/$v$nx.log | cut -c1-40,150-260; head -c 300 $o/$v$nx/bench.log | tail -c 300 | grep -o '"ms_per_step": [0-9.]*'; done
done
