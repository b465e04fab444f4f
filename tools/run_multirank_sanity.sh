#!/bin/bash
# sanity of the multi-rank code path of bench.py on one GPU: dist initialised for one rank; self-halo with config 4's physics
o=gpurun_out/${1:-r5j}; mkdir -p $o
common="--steps 20 --warmup 3 --develop 50 --no-cpu-baseline --no-amr --no-config3 --no-config4 --no-real-mesh"
timeout -k 10 300 python bench.py $common --force-dist --strong-nx 40 > $o/force_dist.json 2> $o/force_dist.err; echo "force-dist rc=$?"; tail -c 700 $o/force_dist.json
timeout -k 10 300 python bench.py $common --self-halo --workload sedov --graph > $o/selfhalo_sedov.json 2> $o/selfhalo_sedov.err; echo "selfhalo sedov rc=$?"; tail -c 900 $o/selfhalo_sedov.json
