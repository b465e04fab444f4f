#!/bin/bash
# rocprofv3 kernel-trace stats of bench.py.  Usage: tools/prof_stats.sh <outdir> [bench args]
out=${1:-gpurun_out/stats}; shift
mkdir -p $out
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
cd "${root:?}" || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-north-star --no-amr --no-config3 --no-config4 --no-real-mesh --develop 0 "$@" > $out/bench.log 2>&1
tail -1 $out/bench.log | cut -c1-400
cat $out/*/*kernel_stats.csv | cut -c1-160 | head -12
