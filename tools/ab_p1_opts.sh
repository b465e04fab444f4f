#!/bin/bash
# A/B of context options on the default library: tools/ab_p1_opts.sh OUT NX ROUNDS "cfg1" "cfg2" ...
out=$1; nx=$2; rounds=$3; shift 3
python tools/ab_p1.py $nx $rounds "$@" >> $out 2>&1
