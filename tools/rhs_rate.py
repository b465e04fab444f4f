#!/usr/bin/env python3
"""Resident RHS/step timing of any order on the synthetic box (for DESIGN.md's
per-order table; bench.py stays on the BASELINE workload).
Usage (GPU box): python tools/rhs_rate.py <ndof> [nx] [limiter] [problem] [timed steps] [dt]
A prescribed dt (config 3: 1e-5 x h / h_fixture) replaces the CFL step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from quinoa_amd import capi, meshgen  # noqa: E402

ndof = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 55
limiter = sys.argv[3] if len(sys.argv) > 3 else "nolimiter"
problem = sys.argv[4] if len(sys.argv) > 4 else "sod_shocktube"
ch = meshgen.kuhn_box(nx, nx, nx)
t1 = time.perf_counter()
kw = dict(bc_extrapolate=[1, 2], bc_sym=[3, 4, 5, 6]) if problem in ("sod_shocktube",) else \
    dict(bc_dirichlet=[1, 2, 3, 4, 5, 6])
dt = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
ctx = capi.Context(ndof, flux="hllc", limiter=limiter, problem=problem, gamma=1.4,
                   cfl=0.0 if dt > 0.0 else 0.3, dt=dt,
                   alpha=0.1, beta=1.0, p0=10.0, **kw)
mesh = capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"])
ctx.synchronize()
t2 = time.perf_counter()
mesh.state_initialize(0.0)
for _ in range(3):
    mesh.step(0.0, want_dt=False)
ctx.synchronize()
mesh.profile_enable(True)
n = int(sys.argv[5]) if len(sys.argv) > 5 else 10
t3 = time.perf_counter()
for _ in range(n):
    mesh.step(0.0, want_dt=False)
ctx.synchronize()
el = (time.perf_counter() - t3) / n
nl, ms = mesh.profile_read()
alg = mesh.rhs_algorithmic_bytes()
print("ndof %d %s %s: %d tets; device mesh build %.2f s; step %.3f ms = %.0f M elem-updates/s; "
      "RHS %.4f ms/launch = %.0f GB/s algorithmic (%.1f %% of 8 TB/s)"
      % (ndof, limiter, problem, mesh.nielem, t2 - t1, el * 1e3, mesh.nielem * 3 / el / 1e6,
         ms / nl, alg / (ms / nl * 1e-3) / 1e9, alg / (ms / nl * 1e-3) / 8e12 * 100))
mesh.close(); ctx.close()
