mkdir -p gpurun_out/r4m
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4m/pytest.log 2>&1; echo "rc $?" >> gpurun_out/r4m/pytest.log; tail -3 gpurun_out/r4m/pytest.log
timeout -k 10 900 python bench.py > gpurun_out/r4m/bench_default.json 2> gpurun_out/r4m/bench_default.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r4m/bench_default.json').read().strip().splitlines()[-1])
print("headline", j["value"], j["ms_per_step"], j["roofline"]["frac"])
ns=j["north_star_point"]; print("ns", ns["value"], ns["ms_per_step"], ns["roofline"]["frac"])
for k in ("at_north_star_size_cold","at_north_star_size"):
    a=j["amr_point"][k]; print(k, a["remesh_total_ms"], a["host_copy_complete_ms"])
print("c3", j["config3_point"]["value"], "c4", j["config4_point"]["value"])
PY
QDG_UPLOAD_STATS=1 timeout -k 10 300 python -c "
import bench, json
print(json.dumps(bench.amr_point(0, nx=119, steps=2, with_partition=False, reserve=True)))" > gpurun_out/r4m/amr119_res.json 2> gpurun_out/r4m/amr119_res.err
tail -12 gpurun_out/r4m/amr119_res.err
