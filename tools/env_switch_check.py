"""All build- and run-time switches of the DG-P1 path (compact / padded task lists, variable tiles,
task-ordered face records, deterministic kernel, tile kernel version 1) on one mesh: four CFL steps
each, maximum difference of the final state against the default configuration.  Expect <= 1e-14.
Usage (GPU box): python tools/env_switch_check.py"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from quinoa_amd import capi, dgmesh, meshgen
ch = meshgen.kuhn_box(12, 11, 10)
chunk = dgmesh.build_chunk(ch["coord"], ch["inpoel"], None, ch["sidesets"])
ref = None
for env in ({}, {"QDG_TASK_COMPACT": "1"}, {"QDG_TILE_TASKS": "512"}, {"QDG_TILE_TASKS": "512", "QDG_TASK_COMPACT": "1"},
            {"QDG_NO_TGEO": "1"}, {"QDG_DETERMINISTIC_RHS": "1"}, {"QDG_TILE_V1": "1"}, {"QDG_TILE_V1": "1", "QDG_TASK_COMPACT": "1"}):
    for k in ("QDG_TASK_COMPACT", "QDG_TILE_TASKS", "QDG_NO_TGEO", "QDG_DETERMINISTIC_RHS", "QDG_TILE_V1"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx = capi.Context(4, flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4, cfl=0.3,
                       bc_extrapolate=[1, 2], bc_sym=[3, 4, 5, 6])
    mesh = dgmesh.upload(ctx, chunk)
    mesh.state_initialize(0.0)
    t = 0.0
    for _ in range(4):
        t += mesh.step(t)
    U = mesh.state_download()
    if ref is None: ref = U
    print(env, "max diff vs default %.2e" % np.abs(U - ref).max(), "t %.6e" % t)
    mesh.close(); ctx.close()
