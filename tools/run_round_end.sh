#!/bin/bash
# round end, in two gpurun calls (each within the 20-minute limit):
#   part 1: full GPU suite, profile set at 1 M tets, one default bench line
#   part 2: profile set at 10.1 M tets, config 3 with counters, self-halo step traces (one / two ghost layers)
tag=${1:-round_end}; part=${2:-1}
o=gpurun_out/$tag; mkdir -p $o
if [ "$part" = 1 ]; then
  python -m pytest tests -m gpu -q > $o/pytest_all.log 2>&1; echo "rc $?" >> $o/pytest_all.log; tail -3 $o/pytest_all.log
  grep -q "^rc 0" $o/pytest_all.log || exit 1
  bash tools/collect_profiles.sh 55 $tag > $o/collect55.log 2>&1; tail -8 $o/collect55.log | cut -c1-200
  timeout -k 10 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err; tail -1 $o/bench_default.json | cut -c1-400
else
  bash tools/collect_profiles.sh 119 $tag > $o/collect119.log 2>&1; tail -8 $o/collect119.log | cut -c1-200
  bash tools/profile_cfg3.sh 110 $o/cfg3_nx110 pmc > $o/cfg3.log 2>&1; tail -4 $o/cfg3.log | cut -c1-300
  bash tools/run_selfhalo_trace.sh $tag/selfhalo_d1 --halo-depth 1 > $o/selfhalo_d1.log 2>&1
  bash tools/run_selfhalo_trace.sh $tag/selfhalo_d2 --halo-depth 2 > $o/selfhalo_d2.log 2>&1
  tail -2 $o/selfhalo_d1.log $o/selfhalo_d2.log
fi
