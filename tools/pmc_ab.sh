#!/bin/bash
# PMC counter groups for the RHS kernels of tools/ab_p1.py configurations (one group per pass;
# --pmc never combined with trace domains other than --kernel-trace).
# Usage: tools/pmc_ab.sh <outdir> <nx> "cfg1" "cfg2" ...      (cfg as for tools/ab_p1.py)
out=$1; nx=$2; shift 2
mkdir -p $out
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
cd "${root:?}" || exit 1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python3 tools/ab_p1.py $nx 1 "$@" > $out/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
res = {k: {c: x / cnt[(k, c)] for c, x in v.items()} for k, v in agg.items() if "k_rhs" in k}
json.dump(res, open("$out/pmc_per_launch.json", "w"), indent=1)
for k, v in sorted(res.items()):
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-44s waves %6d  wave-cyc %6.1fM  wait_any %4.1f%%  wait_inst %4.1f%%  active %4.1f%%  valu-act %4.1f%%  insts_valu %5.1fM  lds insts %5.2fM  lds active %5.1fM  bank-conf %5.1fM  fetch x2 %6.1f MB  write %6.1f MB"
          % (k[-44:], v.get("SQ_WAVES", 0), wc / 1e6, 100 * v.get("SQ_WAIT_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc,
             100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * v.get("SQ_ACTIVE_INST_VALU", 0) / wc, v.get("SQ_INSTS_VALU", 0) / 1e6,
             v.get("SQ_INSTS_LDS", 0) / 1e6, v.get("SQ_ACTIVE_INST_LDS", 0) / 1e6, v.get("SQ_LDS_BANK_CONFLICT", 0) / 1e6,
             2 * v.get("FETCH_SIZE", 0) * 1024 / 1e6, v.get("WRITE_SIZE", 0) * 1024 / 1e6))
PY
