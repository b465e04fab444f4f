#!/bin/bash
# kernel trace of the self-halo step (bench.py --self-halo): what runs between the compute kernels
# usage: tools/run_selfhalo_trace.sh <out dir under gpurun_out> [extra bench flags]
o=gpurun_out/${1:-r5trace}; shift; mkdir -p $o; export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -- python3 bench.py --self-halo --steps 10 --warmup 2 --develop 0 --no-cpu-baseline --no-north-star --no-amr --no-config3 --no-config4 --no-real-mesh "$@" > $o/bench.log 2>&1
tail -1 $o/bench.log | cut -c1-200
f=$(ls $o/trace/*/*kernel_trace.csv | head -1)
python3 tools/trace_step.py $f "k_rhs_p1w<true" -3 > $o/step_timeline.txt; cat $o/step_timeline.txt
cp $(ls $o/trace/*/*kernel_stats.csv | head -1) $o/kernel_stats.csv
rm -rf $o/trace
