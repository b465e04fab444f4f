#!/bin/bash
# WENO experiments: parity of the WENO cases, then config 3 at its size and DG-P1 + WENO at 10.1 M tets, new vs previous kernel
o=gpurun_out/r4ac; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_config3.py tests/test_gpu_parity.py tests/test_gpu_transport.py tests/test_gpu_multirank.py -m gpu -q -x > $o/pytest.log 2>&1; echo "rc $?" >> $o/pytest.log; tail -3 $o/pytest.log
grep -q "rc 0" $o/pytest.log || exit 1
export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then export QDG_LIB=$PWD/quinoa_amd/lib/variants/weno_v2/libqdg.so; fi
  bash tools/profile_cfg3.sh 110 $o/cfg3_$v > $o/cfg3_$v.log 2>&1; echo "== $v P2"; grep "step\|k_weno" $o/cfg3_$v.log | cut -c1-200
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/p1_$v -- python3 tools/rhs_rate.py 4 119 wenop1 vortical_flow 10 1e-6 > $o/p1_$v.log 2>&1; echo "== $v P1"; grep -h "k_weno" $o/p1_$v/*/*kernel_stats.csv | cut -c1-160
done
