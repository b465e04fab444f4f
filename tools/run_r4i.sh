mkdir -p gpurun_out/r4i
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "role_specialised or golden or oracle" > gpurun_out/r4i/pytest.log 2>&1; echo "rc $?" >> gpurun_out/r4i/pytest.log; tail -4 gpurun_out/r4i/pytest.log
timeout -k 10 300 python tools/ab_p1.py 119 3 p1_rhs=0 p1_rhs=2 > gpurun_out/r4i/ab119.log 2>&1; tail -2 gpurun_out/r4i/ab119.log
timeout -k 10 200 python tools/ab_p1.py 55 3 p1_rhs=0 p1_rhs=2 > gpurun_out/r4i/ab55.log 2>&1; tail -2 gpurun_out/r4i/ab55.log
for v in noilp KO_FACE KO_STREAM; do QDG_LIB=$PWD/quinoa_amd/lib/variants/p1r_$v/libqdg.so timeout -k 10 200 python tools/ab_p1.py 119 2 p1_rhs=2 > gpurun_out/r4i/ab119_$v.log 2>&1; echo $v; tail -1 gpurun_out/r4i/ab119_$v.log; done
