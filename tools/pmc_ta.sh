#!/bin/bash
# Texture-addresser / L1 (TA, TCP, TD) counters of the DG-P1 RHS kernel: is the vector-memory front end busy?
# One counter group per pass (--pmc only with --kernel-trace).  Usage: tools/pmc_ta.sh <outdir> <nx>
out=$1; nx=$2
mkdir -p $out
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
cd "${root:?}" || exit 1
rocprofv3 --list-avail > $out/avail.txt 2>&1
pick() { for c in "$@"; do grep -q -w "$c" $out/avail.txt && echo -n "$c "; done; }
g1=$(pick TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE)
g2=$(pick TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum)
g3=$(pick TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum)
g4=$(pick TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum)
g5=$(pick TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_BUSY_avr)
echo "groups: [$g1] [$g2] [$g3] [$g4] [$g5]" > $out/groups.txt
i=0
for grp in "$g1" "$g2" "$g3" "$g4" "$g5"; do
  i=$((i+1))
  [ -z "$grp" ] && continue
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python3 tools/ab_p1.py $nx 1 "p1_rhs=0" > $out/p$i.log 2>&1 || echo "pass $i failed" >> $out/groups.txt
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
res = {k: {c: x / cnt[(k, c)] for c, x in v.items()} for k, v in agg.items() if "k_rhs" in k or "superbee" in k}
json.dump(res, open("$out/pmc_ta_per_launch.json", "w"), indent=1)
for k, v in sorted(res.items()):
    print(k[-50:], {c: round(x / 1e6, 2) for c, x in v.items()})
PY
