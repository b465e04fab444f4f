#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}" || exit 1
o=gpurun_out/r3j; mkdir -p $o
for v in "" _koext _kotasks; do
  for c in FETCH_SIZE WRITE_SIZE; do
    QDG_LIB=$PWD/quinoa_amd/lib/libqdg$v.so timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $o/p${v}_$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-north-star --no-amr --no-config3 --no-config4 > $o/p${v}_$c.log 2>&1 || echo fail $v $c
  done
done
python3 - <<PY
import csv,glob,collections
for v in ["", "_koext", "_kotasks"]:
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob("$o/p%s_*/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0]
            if "k_rhs_p1v" not in k: continue
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k,x in agg.items():
        fs=x["FETCH_SIZE"]/cnt[(k,"FETCH_SIZE")]; ws=x["WRITE_SIZE"]/cnt[(k,"WRITE_SIZE")]
        print("lib%-9s %-34s read %.1f MB write %.1f MB" % (v, k[-30:], 2*fs*1024/1e6, ws*1024/1e6))
PY
