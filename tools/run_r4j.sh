mkdir -p gpurun_out/r4j
timeout -k 10 600 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_partition.py tests/test_gpu_cpp_adapter.py -m gpu -q -x > gpurun_out/r4j/pytest.log 2>&1; echo "rc $?" >> gpurun_out/r4j/pytest.log; tail -4 gpurun_out/r4j/pytest.log
timeout -k 10 300 python bench.py --self-halo --no-cpu-baseline 2> gpurun_out/r4j/selfhalo.err | tail -1 > gpurun_out/r4j/selfhalo_nx55_bench.json
python -c "
import json; j=json.load(open('gpurun_out/r4j/selfhalo_nx55_bench.json')); print('self-halo ms/step', j['ms_per_step'], j['value'])"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-north-star --no-amr --no-config3 --no-config4 2>/dev/null | tail -1 > gpurun_out/r4j/nohalo_nx55_bench.json
python -c "
import json; j=json.load(open('gpurun_out/r4j/nohalo_nx55_bench.json')); print('no-halo ms/step', j['ms_per_step'], j['value'])"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4j/stats -- python3 bench.py --self-halo --no-cpu-baseline > gpurun_out/r4j/stats.log 2>&1
cp gpurun_out/r4j/stats/*/*kernel_stats.csv gpurun_out/r4j/selfhalo_nx55_kernel_stats.csv; cut -c1-110 gpurun_out/r4j/selfhalo_nx55_kernel_stats.csv | head -12
