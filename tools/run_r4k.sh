mkdir -p gpurun_out/r4k
timeout -k 10 900 python bench.py > gpurun_out/r4k/bench_default.json 2> gpurun_out/r4k/bench_default.err
tail -c 300 gpurun_out/r4k/bench_default.json
bash tools/collect_profiles.sh 55 r4k > gpurun_out/r4k/collect55.log 2>&1; tail -5 gpurun_out/r4k/collect55.log
bash tools/collect_profiles.sh 119 r4k > gpurun_out/r4k/collect119.log 2>&1; tail -5 gpurun_out/r4k/collect119.log
