#!/bin/bash
o=gpurun_out/r4y; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_config3.py tests/test_gpu_parity.py -m gpu -q -x -k "weno or config3 or golden" > $o/pytest.log 2>&1; echo "rc $?" >> $o/pytest.log; tail -3 $o/pytest.log
grep -q "rc 0" $o/pytest.log || exit 1
bash tools/profile_cfg3.sh 110 $o/cfg3 > $o/cfg3.log 2>&1; grep "step\|k_weno" $o/cfg3.log | cut -c1-220
QDG_LIB=$PWD/quinoa_amd/lib/variants/weno_w3/libqdg.so bash tools/profile_cfg3.sh 110 $o/cfg3_w3 > $o/cfg3_w3.log 2>&1; grep "step\|k_weno" $o/cfg3_w3.log | cut -c1-220
