"""Per-phase wave timing of the element-centric P1 RHS kernel (k_rhs_p1) from
s_memtime stamps compiled in with -DQDG_STAMPS (debug build, not the product):

  cd quinoa_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DQDG_STAMPS \
      -o ../lib/libqdg_stamps.so qdg_kernels.hip qdg_api.cpp qdg_meshdata.cpp -ldl
  QDG_DETERMINISTIC_RHS=1 python tools/stamps.py        # on the GPU box
"""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, '.')
os.environ["QDG_LIB"] = os.path.abspath("quinoa_amd/lib/libqdg_stamps.so")
from quinoa_amd import capi, dgmesh, meshgen, dg
ch = meshgen.kuhn_box(55, 55, 55)
chunk = dgmesh.build_chunk(ch["coord"], ch["inpoel"], None, ch["sidesets"])
ctx = capi.Context(4, limiter="superbeep1", problem="sod_shocktube", cfl=0.3, bc_extrapolate=[1, 2], bc_sym=[3, 4, 5, 6])
mesh = dgmesh.upload(ctx, chunk); mesh.state_initialize(0.0)
for _ in range(2): mesh.step(0.0, want_dt=False)
ctx.synchronize()
out = np.zeros(16)
capi.lib().qdg_debug_stamps(out.ctypes.data_as(capi.c_f64p), 1)
for _ in range(5): mesh.step(0.0, want_dt=False)
ctx.synchronize()
capi.lib().qdg_debug_stamps(out.ctypes.data_as(capi.c_f64p), 1)
nw = 15 * 15600  # 15 launches x waves
names = ["L1 loads: own row, ids (wait)", "L2 loads: nbr0, geom, coords (wait for coords)", "volume (+src)", "face: wait nbr row/geom + issue prefetch (x4)", "face: 3 GPs compute (x4)", "store R"]
tot = out[:6].sum()
for n, v in zip(names, out[:6]): print("%-52s %10.0f ticks/wave  %5.1f%%" % (n, v / nw, 100 * v / tot))
print("total per wave", tot / nw)
