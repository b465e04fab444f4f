#!/bin/bash
# A/B of variant libraries (tools/build_variant.sh) on the GPU box: tools/ab_libs.sh OUT NX ROUNDS lib1 lib2 ...
# ("default" = quinoa_amd/lib/libqdg.so); every library is timed ROUNDS times, interleaved.
out=$1; nx=$2; rounds=$3; shift 3
for r in $(seq $rounds); do
  for l in "$@"; do
    if [ "$l" = default ]; then unset QDG_LIB; else export QDG_LIB=$PWD/quinoa_amd/lib/variants/$l/libqdg.so; fi
    echo "== lib $l" >> $out
    python tools/ab_p1.py $nx 2 "p1_rhs=0" >> $out 2>&1 || exit 1
  done
done
