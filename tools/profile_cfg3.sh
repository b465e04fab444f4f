#!/bin/bash
# BASELINE config 3 at its own size (vortical_flow DG-P2 + wenop1, 110^3 x 6 = 7 986 000 tets, one
# GPU): timing line + rocprofv3 kernel stats.  Usage: tools/profile_cfg3.sh [nx] [outdir]
nx=${1:-110}
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
cd "${root:?}" || exit 1
out=${2:-gpurun_out/cfg3_nx$nx}
mkdir -p $out
timeout -k 10 600 python3 tools/rhs_rate.py 10 $nx wenop1 vortical_flow > $out/rate.txt 2>&1
cat $out/rate.txt | tail -2
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 tools/rhs_rate.py 10 $nx wenop1 vortical_flow > $out/stats.log 2>&1
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
cut -c1-150 $out/kernel_stats.csv | head -8
