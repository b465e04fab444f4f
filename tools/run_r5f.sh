#!/bin/bash
# DG-P1 experiment (b), bounded first: what would a face record that costs no HBM read buy (knock-out: wrong results,
# valid timing), and what do 240-row tiles (room for a node table in LDS) cost
o=gpurun_out/${1:-r5f}; mkdir -p $o
bash tools/ab_libs.sh $o/ab55.log 55 3 default ko_tgeo tile240 tile240_ko
grep -E "^== lib|RHS median" $o/ab55.log | paste - - | sed 's/nx 55.*RHS median/ RHS median/; s/= .*//'
bash tools/ab_libs.sh $o/ab119.log 119 2 default ko_tgeo tile240 tile240_ko
grep -E "^== lib|RHS median" $o/ab119.log | paste - - | sed 's/nx 119.*RHS median/ RHS median/; s/= .*//'
