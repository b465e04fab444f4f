#!/bin/bash
# BASELINE config 3 (vortical_flow DG-P2 + wenop1; 110^3 x 6 = 7 986 000 tets at its own size, one
# GPU): timing line + rocprofv3 kernel stats, and with a third argument "pmc" the counter passes
# (one group per run, kernel-trace only beside them) summarised per kernel.
# Usage: tools/profile_cfg3.sh [nx] [outdir] [pmc]
nx=${1:-110}
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
cd "${root:?}" || exit 1
out=${2:-gpurun_out/cfg3_nx$nx}
mkdir -p $out
# config 3 prescribes the time step (1e-5, scaled with the mesh size; the 1k-tet
# fixture of the reference case has 10 cells per edge)
dt=$(python3 -c "print(1e-5 * 10 / $nx)")
timeout -k 10 600 python3 tools/rhs_rate.py 10 $nx wenop1 vortical_flow 10 $dt > $out/rate.txt 2>&1
cat $out/rate.txt | tail -2
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 tools/rhs_rate.py 10 $nx wenop1 vortical_flow 10 $dt > $out/stats.log 2>&1
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
cut -c1-150 $out/kernel_stats.csv | head -8
[ "$3" = "pmc" ] || exit 0
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc$i -- python3 tools/rhs_rate.py 10 $nx wenop1 vortical_flow 3 $dt > $out/pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$out/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
res = {k: {c: x / cnt[(k, c)] for c, x in v.items()} for k, v in agg.items()}
for k, v in res.items():
    if "FETCH_SIZE" in v:
        # KiB units; FETCH_SIZE counts 128-B requests as 64 B on gfx950 -> x2 (guide's correction)
        v["hbm_bytes_per_launch"] = (2 * v["FETCH_SIZE"] + v.get("WRITE_SIZE", 0.0)) * 1024
json.dump(res, open("$out/pmc_per_launch.json", "w"), indent=1)
for k, v in res.items():
    if any(s in k for s in ("k_rhs", "k_weno", "k_rk")):
        print(k, {c: round(x) for c, x in v.items()})
PY
