mkdir -p gpurun_out/r4h
QDG_UPLOAD_STATS=1 timeout -k 10 300 python -c "
import bench, json
print(json.dumps(bench.amr_point(0, nx=119, steps=2, with_partition=False, reserve=False)))" > gpurun_out/r4h/amr119_cold.json 2> gpurun_out/r4h/amr119_cold.err
tail -12 gpurun_out/r4h/amr119_cold.err
QDG_UPLOAD_STATS=1 timeout -k 10 300 python -c "
import bench, json
print(json.dumps(bench.amr_point(0, nx=119, steps=2, with_partition=False, reserve=True)))" > gpurun_out/r4h/amr119_res.json 2> gpurun_out/r4h/amr119_res.err
tail -12 gpurun_out/r4h/amr119_res.err
python -c "
import json
for f in ('cold','res'):
    j=json.load(open('gpurun_out/r4h/amr119_%s.json'%f)); print(f, j['rebuild_upload_ms'], j.get('host_copy_complete_ms'), j.get('reserve_ms_outside_the_remesh'))"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_devmesh.py tests/test_gpu_amr.py tests/test_gpu_edge_cases.py -m gpu -q -x > gpurun_out/r4h/pytest.log 2>&1; echo "rc $?" >> gpurun_out/r4h/pytest.log; tail -4 gpurun_out/r4h/pytest.log
timeout -k 10 200 python tools/ab_p1.py 55 3 p1_rhs=0 p1_rhs=2 > gpurun_out/r4h/ab55.log 2>&1; tail -2 gpurun_out/r4h/ab55.log
timeout -k 10 300 python tools/ab_p1.py 119 3 p1_rhs=0 p1_rhs=2 > gpurun_out/r4h/ab119.log 2>&1; tail -2 gpurun_out/r4h/ab119.log
for v in KO_FACE KO_STREAM KO_VOL; do QDG_LIB=$PWD/quinoa_amd/lib/variants/p1r_$v/libqdg.so timeout -k 10 200 python tools/ab_p1.py 119 2 p1_rhs=2 > gpurun_out/r4h/ab119_$v.log 2>&1; echo $v; tail -1 gpurun_out/r4h/ab119_$v.log; done
