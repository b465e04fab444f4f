#!/bin/bash
# the driver's command: default bench.py run; wall time beside it
o=gpurun_out/${1:-r5bench}; mkdir -p $o
SECONDS=0
python bench.py > $o/bench.json 2> $o/bench.err; echo "rc=$? wall=${SECONDS}s"
python3 - <<PY
import json
d=json.loads(open("$o/bench.json").read().strip().splitlines()[-1])
print("headline", round(d["value"]), d["ms_per_step"], d["rates"], d["roofline"]["frac"])
ns=d["north_star_point"]; print("north star", round(ns["value"]), ns["ms_per_step"], ns["rates"], ns["roofline"]["frac"], ns["developed_flow"])
for k in ("config3_point","config4_point"):
    p=d[k]; print(k, round(p["value"]), p["ms_per_step"], p["roofline"]["frac"], p.get("cold_start_M_per_s"))
for p in d["real_mesh_point"]["points"]: print("real", p["tets_total"], round(p["value"]), p["ms_per_step"], p["roofline"]["frac"], p["face_tasks_per_tet"], round(p["cold_start_M_per_s"]))
print(d["config"]["workload"])
PY
