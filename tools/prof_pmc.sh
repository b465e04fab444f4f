#!/bin/bash
# PMC passes for the RHS kernel (one counter group per pass; never combined with
# trace domains other than --kernel-trace).  Usage: tools/prof_pmc.sh <outdir> [bench args]
out=${1:-gpurun_out/pmc}; shift
mkdir -p $out
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
cd "${root:?}" || exit 1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_IFETCH" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-north-star --no-amr --no-config3 --no-config4 --no-real-mesh --develop 0 "$@" > $out/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$out/p*/")):
    for f in glob.glob(d+"**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][-40:]
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
        for k,v in agg.items():
            if "k_rhs" in k or "superbee" in k or "k_rk" in k:
                print(d.split("/")[-2], k, {c: round(x/cnt[(k,c)],1) for c,x in v.items()})
PY
