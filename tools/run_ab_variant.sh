#!/bin/bash
# A/B of a variant library (tools/build_variant.sh NAME ...) against the default one: kernel stats of bench.py at
# 1 M and 10.1 M tets, both in this call.   Usage: tools/run_ab_variant.sh OUTTAG NAME [NAME ...]
o=gpurun_out/$1; mkdir -p $o; shift; names="$@"
for nx in 55 119; do
  extra=""; [ $nx = 119 ] && extra="--nx 119"
  for v in default $names; do
    if [ $v = default ]; then unset QDG_LIB; else export QDG_LIB=$PWD/quinoa_amd/lib/variants/$v/libqdg.so; fi
    bash tools/prof_stats.sh $o/${v}_$nx $extra > $o/${v}_$nx.log 2>&1
    echo "== $v nx $nx: $(grep -o '"ms_per_step": [0-9.]*' $o/${v}_$nx/bench.log | head -1)"
    grep -h "superbee\|k_rhs_p1w" $o/${v}_$nx/*/*kernel_stats.csv | awk -F'",' '{split($2,a,","); printf "   %-44s %10.1f us\n", substr($1,12,44), a[3]/1000}'
  done
done
