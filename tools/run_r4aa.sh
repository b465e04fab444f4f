#!/bin/bash
# round 4 (late): config 4 at full size on one GPU, the warm-started headline, the self-halo step -- after the limiter work
o=gpurun_out/r4aa; mkdir -p $o
F="--no-cpu-baseline --no-north-star --no-amr --no-config3 --no-config4"
timeout -k 10 500 python bench.py --nx 220 --workload sedov --steps 20 --warmup 3 $F > $o/cfg4_full.json 2> $o/cfg4_full.err; tail -1 $o/cfg4_full.json | cut -c1-260
timeout -k 10 300 python bench.py --warmup 1000 $F > $o/warm1000.json 2> $o/warm1000.err; tail -1 $o/warm1000.json | cut -c1-260
timeout -k 10 600 python bench.py --warmup 12000 $F > $o/warm12000.json 2> $o/warm12000.err; tail -1 $o/warm12000.json | cut -c1-260
timeout -k 10 300 python bench.py --self-halo $F > $o/selfhalo.json 2> $o/selfhalo.err; tail -1 $o/selfhalo.json | cut -c1-260
