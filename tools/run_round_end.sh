#!/bin/bash
# round end: full GPU suite, profile sets at 1 M and 10.1 M tets, one default bench line
o=gpurun_out/${1:-round_end}; mkdir -p $o; tag=${1:-round_end}
python -m pytest tests -m gpu -q > $o/pytest_all.log 2>&1; echo "rc $?" >> $o/pytest_all.log; tail -3 $o/pytest_all.log
grep -q "^rc 0" $o/pytest_all.log || exit 1
bash tools/collect_profiles.sh 55 $tag > $o/collect55.log 2>&1; tail -8 $o/collect55.log | cut -c1-200
bash tools/collect_profiles.sh 119 $tag > $o/collect119.log 2>&1; tail -8 $o/collect119.log | cut -c1-200
timeout -k 10 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err; tail -1 $o/bench_default.json | cut -c1-400
