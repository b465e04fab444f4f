#!/usr/bin/env python3
"""The device re-mesh of a rank's chunk (qdg_mesh_refine_chunk) against the host path (qdg_refine_chunk ->
qdg_mesh_from_chunk_gid -> qdg_state_transfer) at a REAL size: the nx^3 Kuhn box in NPARTS RCB chunks with ghost
halos, all on one GPU; Sod DG-P1 + Superbee (reproducible kernel), 2 steps, re-mesh both ways, 2 more steps.
Compares plan and numbering exactly and the states bitwise.   Usage: python tools/remesh_at_size.py NX NPARTS"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from quinoa_amd import amr, capi, dg, meshgen, partition  # noqa: E402

nx, nparts = int(sys.argv[1]), int(sys.argv[2])
kw = dict(flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4, cfl=0.3,
          bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])
opt = {"keep_connectivity": 1, "p1_rhs": 1}
g = meshgen.kuhn_box(nx, nx, nx)
part = partition.partition(g["coord"], g["inpoel"], nparts, "rcb")
chunks = [partition.build_chunk(g["coord"], g["inpoel"], g["sidesets"], part, nparts, r) for r in range(nparts)]
ctxa, ctxb = capi.Context(4, options=opt, **kw), capi.Context(4, options=opt, **kw)


def build(ctx, ch):
    return capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"], nielem=ch["nielem"],
                                       elem_gid=ch["gid"])


A, B = [build(ctxa, c) for c in chunks], [build(ctxb, c) for c in chunks]
for m in A + B:
    m.state_initialize(0.0)
da, db = dg.LocalChunks(ctxa, A, chunks), dg.LocalChunks(ctxb, B, chunks)
t = tb = 0.0
for _ in range(2):
    t += da.step(t); tb += db.step(tb)
assert t == tb
A2, B2, chA, chB = [], [], [], []
th = td = 0.0
for ch, ma, mb in zip(chunks, A, B):
    t0 = time.perf_counter()
    ch2, par = amr.refine_chunk(ch)
    m2 = build(ctxa, ch2)
    amr.state_transfer(ma, m2, par)
    ctxa.synchronize()
    t1 = time.perf_counter()
    n2, plan = mb.refine_chunk(ch["nbr_rank"])
    ctxb.synchronize()
    t2 = time.perf_counter()
    th, td = max(th, t1 - t0), max(td, t2 - t1)
    assert plan["nielem"] == ch2["nielem"] and np.array_equal(plan["gid"], ch2["gid"])
    assert np.array_equal(plan["parent"], par) and plan["recv_counts"] == list(ch2["recv_counts"])
    assert all(np.array_equal(p, q) for p, q in zip(plan["send_lists"], ch2["send_lists"]))
    ma.close(); mb.close()
    A2.append(m2); B2.append(n2); chA.append(ch2); chB.append(plan)
da, db = dg.LocalChunks(ctxa, A2, chA), dg.LocalChunks(ctxb, B2, chB)
for _ in range(2):
    t += da.step(t); tb += db.step(tb)
assert t == tb
same = all(np.array_equal(ma.state_download(), mb.state_download()) for ma, mb in zip(A2, B2))
print("nx %d (%d tets) in %d chunks: %s owned -> %s owned + %s ghost tets per chunk; plan, numbering and parents "
      "identical; states after 2 + 2 steps bitwise equal: %s; re-mesh per rank: host path %.0f ms, device %.1f ms"
      % (nx, 6 * nx ** 3, nparts, [c["nielem"] for c in chunks], [c["nielem"] for c in chB],
         [len(c["gid"]) - c["nielem"] for c in chB], same, th * 1e3, td * 1e3), flush=True)
for m in A2 + B2:
    m.close()
ctxa.close(); ctxb.close()
sys.exit(0 if same else 1)
