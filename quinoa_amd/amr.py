"""Mesh refinement during time stepping on the resident path (BASELINE config 5): uniform
1:8 refinement on the host (qdg_refine_uniform), mesh-derived data of the new mesh on the
device (qdg_mesh_from_connectivity -> qdg_dev_facedata), the state carried over on the device
(qdg_state_transfer: a child takes its parent's row, DG::resizePostAMR, DG.cpp:1597-1605)."""
import ctypes as C
import time

import numpy as np

from . import capi


def refine_uniform(coord, inpoel, sidesets):
    """-> coord2[nnode2,3], inpoel2[8*ne,4], sidesets2 {id: tri[4*n,3]}, parent[8*ne]"""
    L = capi.lib()
    coord = np.asarray(coord, dtype=np.float64)
    inp, pinp = capi._sz(np.asarray(inpoel).reshape(-1))
    ne, nn = len(inp) // 4, coord.shape[0]
    x, px = capi._f64(coord[:, 0]); y, py = capi._f64(coord[:, 1]); z, pz = capi._f64(coord[:, 2])
    ids = sorted(sidesets or {})
    tri = np.concatenate([np.asarray(sidesets[s]).reshape(-1, 3) for s in ids]) if ids else np.zeros((0, 3))
    tset = np.concatenate([np.full(len(sidesets[s]), s, np.int64) for s in ids]) if ids else np.zeros(0, np.int64)
    tri, ptri = capi._sz(tri.reshape(-1))
    h = C.c_void_p()
    capi._chk(L.qdg_refine_uniform(C.c_size_t(ne), C.c_size_t(nn), pinp, px, py, pz, C.c_size_t(len(tset)),
                                   ptri, C.byref(h)))
    try:
        n2 = C.c_size_t()
        capi._chk(L.qdg_refined_get(h, C.byref(n2), None, None, None, None, None, None))
        n2 = int(n2.value)
        inp2 = np.zeros(32 * ne, dtype=np.uint64)
        par = np.zeros(8 * ne, dtype=np.uint64)
        c2 = np.zeros((3, n2))
        tri2 = np.zeros(max(1, 12 * len(tset)), dtype=np.uint64)
        capi._chk(L.qdg_refined_get(h, None, inp2.ctypes.data_as(capi.c_szp), par.ctypes.data_as(capi.c_szp),
                                    c2[0].ctypes.data_as(capi.c_f64p), c2[1].ctypes.data_as(capi.c_f64p),
                                    c2[2].ctypes.data_as(capi.c_f64p), tri2.ctypes.data_as(capi.c_szp)))
    finally:
        L.qdg_refined_destroy(h)
    tri2 = tri2[:12 * len(tset)].astype(np.int64).reshape(-1, 3)
    tset2 = np.repeat(tset, 4)
    ss2 = {int(s): tri2[tset2 == s] for s in ids}
    return np.ascontiguousarray(c2.T), inp2.astype(np.int64).reshape(-1, 4), ss2, par.astype(np.int64)


def state_transfer(mesh_from, mesh_to, parent):
    par, ppar = capi._sz(parent)
    capi._chk(capi.lib().qdg_state_transfer(mesh_from.h, mesh_to.h, ppar))


class RefinedRun:
    """One chunk without ghosts that is refined uniformly while it runs: holds the host mesh
    (what Discretization holds), the device mesh handle and the resident state."""

    def __init__(self, ctx, coord, inpoel, sidesets):
        self.ctx = ctx
        self.coord, self.inpoel, self.sidesets = np.asarray(coord, dtype=np.float64), np.asarray(inpoel), sidesets
        self.mesh = capi.mesh_from_connectivity(ctx, self.inpoel, self.coord, self.sidesets)
        self.timings = []

    def refine(self):
        """uniform 1:8 refinement + rebuild on the device + state transfer; returns the seconds
        spent in (host refinement, device mesh rebuild incl. upload, state transfer)"""
        t0 = time.perf_counter()
        c2, i2, s2, par = refine_uniform(self.coord, self.inpoel, self.sidesets)
        t1 = time.perf_counter()
        new = capi.mesh_from_connectivity(self.ctx, i2, c2, s2)
        self.ctx.synchronize()
        t2 = time.perf_counter()
        state_transfer(self.mesh, new, par)
        t3 = time.perf_counter()
        self.mesh.close()
        self.mesh, self.coord, self.inpoel, self.sidesets = new, c2, i2, s2
        self.timings.append((t1 - t0, t2 - t1, t3 - t2))
        return self.timings[-1]
