#!/usr/bin/env python3
"""Can the bandwidth-bound limiter pass run BESIDE the DG-P1 RHS kernel?  Two independent meshes on one GPU,
each under its own context (own stream): N fused-RK RHS launches on mesh A, N Superbee passes on mesh B,
first one after the other, then enqueued interleaved on the two streams.  If the two kernels share the chip
well the interleaved time approaches max(T_rhs, T_lim); if the dispatcher serialises them it stays at the sum.
Usage: python tools/coschedule_probe.py [NX] [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinoa_amd import capi, meshgen  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 95
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ch = meshgen.kuhn_box(nx, nx, nx)
kw = dict(flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4, dt=1e-6,
          bc_extrapolate=[1, 2], bc_sym=[3, 4, 5, 6])
ctxs = [capi.Context(4, options={"fused_update": 0}, **kw) for _ in range(2)]
meshes = [capi.mesh_from_connectivity(c, ch["inpoel"], ch["coord"], ch["sidesets"]) for c in ctxs]
for m in meshes:
    m.state_initialize(0.0)
    m.step(0.0, want_dt=False)
for c in ctxs:
    c.synchronize()
A, B = meshes


def sync():
    for c in ctxs:
        c.synchronize()


def rhs(m):          # the three RHS launches of a prescribed-dt step (RK update fused in), no limiter
    for s_ in range(3):
        m.stage_rhs_dt(s_, 0.0)
        m.stage_update(s_)


def lim(m):          # the three limiter passes of a step
    for _ in range(3):
        m.stage_limit()


def run(fa, fb):
    sync()
    t0 = time.perf_counter()
    for _ in range(N):
        if fa:
            fa(A)
        if fb:
            fb(B)
    sync()
    return (time.perf_counter() - t0) / N * 1e3


res = {}
for rep in range(3):
    res.setdefault("rhs alone", []).append(run(rhs, None))
    res.setdefault("limiter alone", []).append(run(None, lim))
    res.setdefault("both, two streams", []).append(run(rhs, lim))
    res.setdefault("rhs + rhs, two streams", []).append(run(rhs, rhs))
ne = A.nielem
for k, v in res.items():
    print("nx %d (%d tets per mesh)  %-26s %.3f ms per iteration (min of %d)" % (nx, ne, k, min(v), len(v)), flush=True)
a, b, c = min(res["rhs alone"]), min(res["limiter alone"]), min(res["both, two streams"])
print("sum %.3f  max %.3f  measured together %.3f  -> %.0f %% of the sum" % (a + b, max(a, b), c, 100 * c / (a + b)))
for m in meshes:
    m.close()
for c in ctxs:
    c.close()
