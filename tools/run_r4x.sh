#!/bin/bash
# round 4: WENO with a lean lane (gradient modes only) and a patching row copy -- parity, then config 3 at its size
o=gpurun_out/r4x; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config3.py tests/test_gpu_transport.py tests/test_gpu_edge_cases.py tests/test_gpu_multirank.py tests/test_gpu_amr.py -m gpu -q -x > $o/pytest.log 2>&1; echo "rc $?" >> $o/pytest.log; tail -3 $o/pytest.log
grep -q "rc 0" $o/pytest.log || exit 1
bash tools/profile_cfg3.sh 110 $o/cfg3 > $o/cfg3.log 2>&1; tail -12 $o/cfg3.log | cut -c1-220
