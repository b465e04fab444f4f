#!/bin/bash
# Diagnostic build of libqdg with s_memtime stamps in k_rhs_p1w (-DQDG_P1_STAMPS) and a run that
# prints where a wave of the DG-P1 tile kernel spends its cycles.  Never a timed build: the
# stamps' fences forbid overlaps the real kernel has -- read the SHARES.
# Usage: tools/stamps_p1.sh NX VARIANT     (on the GPU box; builds into /tmp)
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$root/quinoa_amd/csrc" || exit 1
out=/tmp/qdg_stamps; mkdir -p $out
for f in qdg_rhs_p1.hip qdg_rhs_p2.hip qdg_kernels.hip qdg_devmesh.hip qdg_api.cpp qdg_meshdata.cpp qdg_partition.cpp qdg_exo.cpp; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DQDG_P1_STAMPS -c $f -o $out/${f%.*}.o &
done
wait
hipcc --offload-arch=gfx950 -fPIC -shared -o $out/libqdg_stamps.so $out/*.o -ldl || exit 1
cd "$root" && QDG_LIB=$out/libqdg_stamps.so python3 - "$@" <<'PY'
import sys, ctypes as C
import numpy as np
sys.path.insert(0, '.')
from quinoa_amd import capi, meshgen
nx, variant = int(sys.argv[1]), int(sys.argv[2])
ch = meshgen.kuhn_box(nx, nx, nx)
ctx = capi.Context(4, flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4, cfl=0.3,
                   bc_extrapolate=[1, 2], bc_sym=[3, 4, 5, 6], options={"p1_variant": variant})
mesh = capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"])
mesh.state_initialize(0.0)
for _ in range(3):
    mesh.step(0.0, want_dt=False)
ctx.synchronize()
out = np.zeros(32)
capi.lib().qdg_debug_stamps(out.ctypes.data_as(capi.c_f64p), 1)
for _ in range(5):
    mesh.step(0.0, want_dt=False)
ctx.synchronize()
capi.lib().qdg_debug_stamps(out.ctypes.data_as(capi.c_f64p), 1)
names = ["phase 0 (loads, nodal, LDS)", "barrier 1", "round 0", "round 1", "round 2", "round 3", "phase-2 requests",
         "barrier 2", "phase 2 compute", "barrier 3", "staging + stores"]
for grp, off in (("waves 0-1", 0), ("waves 2+", 16)):
    tot = out[off:off + 11].sum()
    print(grp, "total %.3e cycles" % tot)
    for i, n in enumerate(names):
        print("   %-30s %5.1f %%" % (n, 100 * out[off + i] / tot))
PY
