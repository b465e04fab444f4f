#!/bin/bash
# A/B of variant libraries built with other compiler flags (tools/build_variant.sh): DG-P1 RHS at nx^3 (tools/ab_p1.py)
# and config 3's DG-P2 RHS (tools/rhs_rate.py).   Usage: tools/ab_flags.sh OUT NX1 NX2 lib1 lib2 ...  ("default" = the product build)
out=$1; nx1=$2; nx2=$3; shift 3
: > $out
dt=$(python3 -c "print(1e-5 * 10 / $nx2)")
for r in 1 2; do
  for l in "$@"; do
    if [ "$l" = default ]; then unset QDG_LIB; else export QDG_LIB=$PWD/quinoa_amd/lib/variants/$l/libqdg.so; fi
    echo "== lib $l (round $r)" >> $out
    python tools/ab_p1.py $nx1 2 "p1_rhs=0" 2>&1 | tail -1 >> $out
    python tools/rhs_rate.py 10 $nx2 wenop1 vortical_flow 10 $dt 2>&1 | tail -1 >> $out
  done
done
cat $out
