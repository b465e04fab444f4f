#!/bin/bash
# config 3 at its size: default library vs a variant.  Usage: tools/run_ab_cfg3.sh OUTTAG NAME
o=gpurun_out/$1; mkdir -p $o; name=$2
for v in default $name; do
  if [ $v = default ]; then unset QDG_LIB; else export QDG_LIB=$PWD/quinoa_amd/lib/variants/$v/libqdg.so; fi
  bash tools/profile_cfg3.sh 110 $o/cfg3_$v > $o/cfg3_$v.log 2>&1; echo "== $v"; grep "step\|k_weno\|k_rhs_p2s" $o/cfg3_$v.log | cut -c1-60,150-240
done
