#!/bin/bash
# round 5, first GPU pass: GPU tests after the clean-up, then the self-halo step with plain launches and as a hipGraph
o=gpurun_out/${1:-r5a}; mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $o/pytest.log; tail -3 $o/pytest.log
common="--self-halo --steps 50 --warmup 5 --no-cpu-baseline --no-north-star --no-amr --no-config3 --no-config4 --no-real-mesh"
timeout -k 10 300 python bench.py $common --no-graph --develop 0 > $o/selfhalo_plain.json 2> $o/selfhalo_plain.err && tail -c 600 $o/selfhalo_plain.json
timeout -k 10 300 python bench.py $common --develop 0 > $o/selfhalo_graph.json 2> $o/selfhalo_graph.err && tail -c 600 $o/selfhalo_graph.json
