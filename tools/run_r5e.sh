#!/bin/bash
# DG-P1 experiment (a): bank-aware face-task order; timings + SQ_LDS_BANK_CONFLICT per order
o=gpurun_out/${1:-r5e}; mkdir -p $o; export TMPDIR=/tmp
python tools/ab_taskorder.py 55 5 > $o/ab55.log 2>&1; tail -3 $o/ab55.log
python tools/ab_taskorder.py 119 3 > $o/ab119.log 2>&1; tail -3 $o/ab119.log
for ord in 0 1 2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $o/pmc$ord -- python3 tools/ab_taskorder.py 55 1 $ord > $o/pmc$ord.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$o/pmc$ord/*/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0]
    if "k_rhs_p1w" not in k: continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    if r["Counter_Name"]=="SQ_WAVE_CYCLES": cnt[k]+=1
for k in acc: print("order $ord", k[-40:], {c: round(v/max(cnt[k],1)/1e6,2) for c,v in acc[k].items()}, "launches", cnt[k])
PY
done
rm -rf $o/pmc0 $o/pmc1 $o/pmc2
