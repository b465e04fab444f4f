#!/usr/bin/env python3
"""Interleaved A/B timing of DG-P1 RHS kernel forms on ONE resident mesh (GPU box).
Usage: python tools/ab_p1.py NX [ROUNDS] "p1_rhs=0" "p1_rhs=1" "fused_update=0" ...
Every configuration is a comma-separated list of qdg_ctx_set_option settings applied to the
same context (the options are read at launch time); ROUNDS interleaved rounds of 3 + 10 steps
of the Sod DG-P1 + Superbee workload each; RHS kernel time from the library's event pairs;
prints the median and the minimum per configuration and the deviation of the final state of
the first round from the first configuration's."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from quinoa_amd import capi, meshgen  # noqa: E402

nx = int(sys.argv[1])
args = sys.argv[2:]
rounds = 3
if args and args[0].isdigit():
    rounds = int(args[0]); args = args[1:]
cfgs = args or ["p1_rhs=0"]
BASE = {"p1_rhs": 0, "fused_update": 1}

ch = meshgen.kuhn_box(nx, nx, nx)
ctx = capi.Context(4, flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4, cfl=0.3,
                   bc_extrapolate=[1, 2], bc_sym=[3, 4, 5, 6])
mesh = capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"])
ne = mesh.nielem
res = {c: [] for c in cfgs}
step = {c: [] for c in cfgs}
ref = None
dev = {}
for r in range(rounds):
    for cfg in cfgs:
        for k, v in BASE.items():
            ctx.set_option(k, v)
        for kv in cfg.split(","):
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
        mesh.state_initialize(0.0)
        for _ in range(3):
            mesh.step(0.0, want_dt=False)
        ctx.synchronize()
        mesh.profile_enable(True)
        n = 10
        t3 = time.perf_counter()
        for _ in range(n):
            mesh.step(0.0, want_dt=False)
        ctx.synchronize()
        el = (time.perf_counter() - t3) / n
        nl, ms = mesh.profile_read()
        mesh.profile_enable(False)
        res[cfg].append(ms / nl); step[cfg].append(el * 1e3)
        if r == 0:
            U = mesh.state_download()
            if ref is None:
                ref = U
            dev[cfg] = np.abs(U - ref).max() / np.abs(ref).max()
alg = mesh.rhs_algorithmic_bytes()
for cfg in cfgs:
    med, mn = float(np.median(res[cfg])), float(np.min(res[cfg]))
    print("nx %d (%d tets) %-28s RHS median %.4f min %.4f ms/launch = %.1f %% of 8 TB/s (median)  step %.3f ms  "
          "|U - U_first| %.1e" % (nx, ne, cfg, med, mn, alg / (med * 1e-3) / 8e12 * 100,
                                  float(np.median(step[cfg])), dev[cfg]), flush=True)
mesh.close(); ctx.close()
