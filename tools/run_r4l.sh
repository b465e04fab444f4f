mkdir -p gpurun_out/r4l
QDG_UPLOAD_STATS=1 timeout -k 10 300 python -c "
import bench, json
print(json.dumps(bench.amr_point(0, nx=119, steps=2, with_partition=False, reserve=False)))" > gpurun_out/r4l/amr119_cold.json 2> gpurun_out/r4l/amr119_cold.err
tail -12 gpurun_out/r4l/amr119_cold.err
python -c "
import json
j=json.load(open('gpurun_out/r4l/amr119_cold.json')); print('cold fresh box', j['remesh_total_ms'], j['host_copy_complete_ms'])"
timeout -k 10 600 python -m pytest tests/test_gpu_devmesh.py tests/test_gpu_amr.py tests/test_gpu_partition.py tests/test_gpu_edge_cases.py -m gpu -q -x > gpurun_out/r4l/pytest.log 2>&1; echo "rc $?" >> gpurun_out/r4l/pytest.log; tail -3 gpurun_out/r4l/pytest.log
bash tools/profile_cfg3.sh 110 gpurun_out/r4l/cfg3 > gpurun_out/r4l/cfg3.log 2>&1; tail -6 gpurun_out/r4l/cfg3.log
