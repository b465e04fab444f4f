#!/usr/bin/env python3
"""PCIe-inclusive rate of the stateless, DGPDE-shaped entry point qdg_rhs (host
U in, host R out) on the bench workload -- quoted in DESIGN.md §6, never the
bench `value`.   Usage (GPU box): python tools/pcie_rate.py [nx]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from quinoa_amd import capi, dgmesh, meshgen  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 55
ch = meshgen.kuhn_box(nx, nx, nx)
chunk = dgmesh.build_chunk(ch["coord"], ch["inpoel"], None, ch["sidesets"])
ctx = capi.Context(4, flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4,
                   cfl=0.3, bc_extrapolate=[1, 2], bc_sym=[3, 4, 5, 6])
mesh = dgmesh.upload(ctx, chunk)
U = mesh.initialize(0.0)
for _ in range(2):
    R = mesh.rhs(0.0, U)
n = 5
t0 = time.perf_counter()
for _ in range(n):
    R = mesh.rhs(0.0, U)
el = (time.perf_counter() - t0) / n
print("qdg_rhs host->host: %d tets, %.2f ms per call, %.1f M element-updates/s (U %.0f MB in, R %.0f MB out)"
      % (chunk.nielem, el * 1e3, chunk.nielem / el / 1e6, U.nbytes / 1e6, R.nbytes / 1e6))
mesh.close(); ctx.close()
