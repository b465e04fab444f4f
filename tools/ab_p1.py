#!/usr/bin/env python3
"""A/B timing of the DG-P1 step variants on one mesh build (GPU box).
Usage: python tools/ab_p1.py NX "ENV1=a,ENV2=b" "ENV1=c" ...   ('-' = defaults)
Each configuration: fresh upload (the tile layout is chosen at upload), 3 + 10 steps of the
Sod DG-P1 + Superbee workload, RHS kernel time from the library's event pairs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401
from quinoa_amd import capi, dgmesh, meshgen  # noqa: E402

nx = int(sys.argv[1])
ch = meshgen.kuhn_box(nx, nx, nx)
chunk = dgmesh.build_chunk(ch["coord"], ch["inpoel"], None, ch["sidesets"])
ref = None
for cfg in sys.argv[2:]:
    added = []
    if cfg != "-":
        for kv in cfg.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
            added.append(k)
    ctx = capi.Context(4, flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4, cfl=0.3,
                       bc_extrapolate=[1, 2], bc_sym=[3, 4, 5, 6])
    mesh = dgmesh.upload(ctx, chunk)
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
    alg = mesh.rhs_algorithmic_bytes()
    U = mesh.state_download()
    if ref is None:
        ref = U
    dev = np.abs(U - ref).max() / np.abs(ref).max()
    print("nx %d %-44s step %.3f ms  RHS %.4f ms/launch  %.1f %% of 8 TB/s   |U - U_first| %.1e"
          % (nx, cfg, el * 1e3, ms / nl, alg / (ms / nl * 1e-3) / 8e12 * 100, dev), flush=True)
    mesh.close(); ctx.close()
    for k in added:
        del os.environ[k]
