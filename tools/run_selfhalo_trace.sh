#!/bin/bash
# kernel trace of the self-halo step (bench.py --self-halo): what runs between the compute kernels
o=gpurun_out/${1:-r4am}; mkdir -p $o; export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $o/trace -- python3 bench.py --self-halo --steps 10 --warmup 2 --no-cpu-baseline --no-north-star --no-amr --no-config3 --no-config4 > $o/bench.log 2>&1
tail -1 $o/bench.log | cut -c1-200
ls $o/trace/*/
