import sys, time, os
sys.path.insert(0, '.')
import numpy as np
from quinoa_amd import amr, capi, meshgen
ch = meshgen.kuhn_box(32, 32, 32)
ctx = capi.Context(4, flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4, cfl=0.3, bc_extrapolate=[1, 2], bc_sym=[3, 4, 5, 6])
run = amr.RefinedRun(ctx, ch["coord"], ch["inpoel"], ch["sidesets"])
run.mesh.state_initialize(0.0)
for k in range(2):
    sys.stderr.write("---- refinement %d\n" % k)
    print(run.refine(), run.mesh.nielem)
    if k == 0:
        # back to the small mesh for a second, warmed-up measurement
        run.mesh.close()
        run = amr.RefinedRun(ctx, ch["coord"], ch["inpoel"], ch["sidesets"]); run.mesh.state_initialize(0.0)
