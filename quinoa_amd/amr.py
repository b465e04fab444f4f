"""Mesh refinement during time stepping on the resident path (BASELINE config 5): uniform
1:8 refinement on the host (qdg_refine_uniform), mesh-derived data of the new mesh on the
device (qdg_mesh_from_connectivity -> qdg_dev_facedata), the state carried over on the device
(qdg_state_transfer: a child takes its parent's row, DG::resizePostAMR, DG.cpp:1597-1605)."""
import ctypes as C
import time

import numpy as np

from . import capi


def refine_uniform(coord, inpoel, sidesets, ctx=None):
    """-> coord2[nnode2,3], inpoel2[8*ne,4], sidesets2 {id: tri[4*n,3]}, parent[8*ne]
    ctx: compute on that context's GPU (qdg_refine_uniform_device: same arrays)"""
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
    if ctx is not None:
        capi._chk(L.qdg_refine_uniform_device(ctx.h, C.c_size_t(ne), C.c_size_t(nn), pinp, px, py, pz,
                                              C.c_size_t(len(tset)), ptri, C.byref(h)))
    else:
        capi._chk(L.qdg_refine_uniform(C.c_size_t(ne), C.c_size_t(nn), pinp, px, py, pz, C.c_size_t(len(tset)),
                                       ptri, C.byref(h)))
    try:
        n2 = C.c_size_t()
        capi._chk(L.qdg_refined_get(h, C.byref(n2), None, None, None, None, None, None))
        n2 = int(n2.value)
        inp2 = np.empty(32 * ne, dtype=np.uint64)          # (filled by the library)
        par = np.empty(8 * ne, dtype=np.uint64)
        c2 = np.empty((3, n2))
        tri2 = np.zeros(max(1, 12 * len(tset)), dtype=np.uint64)
        capi._chk(L.qdg_refined_get(h, None, inp2.ctypes.data_as(capi.c_szp), par.ctypes.data_as(capi.c_szp),
                                    c2[0].ctypes.data_as(capi.c_f64p), c2[1].ctypes.data_as(capi.c_f64p),
                                    c2[2].ctypes.data_as(capi.c_f64p), tri2.ctypes.data_as(capi.c_szp)))
    finally:
        L.qdg_refined_destroy(h)
    tri2 = tri2[:12 * len(tset)].view(np.int64).reshape(-1, 3)
    tset2 = np.repeat(tset, 4)
    ss2 = {int(s): tri2[tset2 == s] for s in ids}
    return np.ascontiguousarray(c2.T), inp2.view(np.int64).reshape(-1, 4), ss2, par.view(np.int64)


def derefine_uniform(coord, inpoel, sidesets):
    """qdg_derefine_uniform: the inverse of refine_uniform for a mesh in its order (children 8 e + k, old nodes first,
    child triangles 4 t + k per side set) -> coord[nnode_old, 3], inpoel[ne / 8, 4], sidesets {id: tri[n / 4, 3]}"""
    L = capi.lib()
    coord = np.asarray(coord, dtype=np.float64)
    inp, pinp = capi._sz(np.asarray(inpoel).reshape(-1))
    ne, nn = len(inp) // 4, coord.shape[0]
    x, px = capi._f64(coord[:, 0]); y, py = capi._f64(coord[:, 1]); z, pz = capi._f64(coord[:, 2])
    ids = sorted(sidesets or {})
    tri = np.concatenate([np.asarray(sidesets[s]).reshape(-1, 3) for s in ids]) if ids else np.zeros((0, 3))
    tset = np.concatenate([np.full(len(sidesets[s]), s, np.int64) for s in ids]) if ids else np.zeros(0, np.int64)
    tri, ptri = capi._sz(tri.reshape(-1) if len(tset) else np.zeros(3))
    h = C.c_void_p()
    capi._chk(L.qdg_derefine_uniform(C.c_size_t(ne), C.c_size_t(nn), pinp, px, py, pz, C.c_size_t(len(tset)), ptri,
                                     C.byref(h)))
    try:
        n2 = C.c_size_t()
        capi._chk(L.qdg_refined_get(h, C.byref(n2), None, None, None, None, None, None))
        n2 = int(n2.value)
        inp2 = np.empty(ne // 2, dtype=np.uint64)
        c2 = np.empty((3, n2))
        tri2 = np.zeros(max(1, 3 * (len(tset) // 4)), dtype=np.uint64)
        capi._chk(L.qdg_refined_get(h, None, inp2.ctypes.data_as(capi.c_szp), None, c2[0].ctypes.data_as(capi.c_f64p),
                                    c2[1].ctypes.data_as(capi.c_f64p), c2[2].ctypes.data_as(capi.c_f64p),
                                    tri2.ctypes.data_as(capi.c_szp)))
    finally:
        L.qdg_refined_destroy(h)
    tri2 = tri2[:3 * (len(tset) // 4)].view(np.int64).reshape(-1, 3)
    tset2 = tset[::4]
    return np.ascontiguousarray(c2.T), inp2.view(np.int64).reshape(-1, 4), {int(s): tri2[tset2 == s] for s in ids}


def state_transfer(mesh_from, mesh_to, parent):
    par, ppar = capi._sz(parent)
    capi._chk(capi.lib().qdg_state_transfer(mesh_from.h, mesh_to.h, ppar))


def state_migrate(mesh_from, gid_from, mesh_to, gid_to):
    """qdg_state_migrate: owned rows of `mesh_to` whose global tet id is owned by `mesh_from` are
    copied device to device (both chunks under one context); returns the number of rows moved"""
    gf, pgf = capi._sz(np.asarray(gid_from))
    gt, pgt = capi._sz(np.asarray(gid_to))
    n = C.c_size_t()
    capi._chk(capi.lib().qdg_state_migrate(mesh_from.h, pgf, mesh_to.h, pgt, C.byref(n)))
    return n.value


class RefinedRun:
    """One chunk without ghosts that is refined uniformly while it runs: holds the host mesh
    (what Discretization holds), the device mesh handle and the resident state."""

    def __init__(self, ctx, coord, inpoel, sidesets, device_refine=True, resident=False):
        """resident: the re-mesh never leaves the device (qdg_mesh_refine_uniform; the mesh handle keeps its
        connectivity resident, the host gets its copy of the refined mesh from a second thread)"""
        self.ctx = ctx
        self.device_refine = device_refine
        self.resident = resident
        self.coord, self.inpoel, self.sidesets = np.asarray(coord, dtype=np.float64), np.asarray(inpoel), sidesets
        # keep_connectivity concerns THIS run's meshes only: the caller's context gets its setting back (a
        # re-meshed handle inherits the kept arrays from its parent handle, not from the option)
        keep0 = ctx.get_option("keep_connectivity") if resident else None
        if resident:
            ctx.set_option("keep_connectivity", 1)
        try:
            self.mesh = capi.mesh_from_connectivity(ctx, self.inpoel, self.coord, self.sidesets)
        finally:
            if resident:
                ctx.set_option("keep_connectivity", keep0)
        self.timings = []
        self.host_copy_s = None

    def refine(self):
        """uniform 1:8 refinement (on the device by default, copied back for the host's book-keeping)
        + rebuild on the device + state transfer; returns the seconds spent in (refinement, device
        mesh rebuild incl. upload, state transfer).  resident: (0, the one call that does all three, 0);
        self.host_copy_s = seconds from the start of that call until the host holds the refined mesh."""
        if self.resident:
            t0 = time.perf_counter()
            new, ref = self.mesh.refine_uniform(host_copy=True)
            self.ctx.synchronize()
            t1 = time.perf_counter()
            self.mesh.close()
            self.mesh = new
            self.coord, self.inpoel, self.sidesets, _ = ref.get()       # waits for the copying thread
            self.host_copy_s = time.perf_counter() - t0
            ref.close()
            self.timings.append((0.0, t1 - t0, 0.0))
            return self.timings[-1]
        t0 = time.perf_counter()
        c2, i2, s2, par = refine_uniform(self.coord, self.inpoel, self.sidesets,
                                         ctx=self.ctx if self.device_refine else None)
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


def _face_keys(inpoel):
    """sorted node triples of the 4 faces of every tet: [ne*4, 3], row 4*e + lf"""
    lp = np.array([[1, 2, 3], [2, 0, 3], [3, 0, 1], [0, 2, 1]])
    return np.sort(np.asarray(inpoel)[:, lp].reshape(-1, 3), axis=1)


def _rows(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a.view([("", a.dtype)] * a.shape[1]).reshape(-1)


def refine_chunk(ch):
    """qdg_refine_chunk (C++, all host cores): uniform 1:8 refinement of ONE RANK's chunk of a
    partitioned mesh with its new ghost layer and halo plan; see _refine_chunk_numpy for the
    statement of the algorithm (kept as the cross-check of the tests: about 400x slower).
    Returns (new chunk dict, parent[nunk_new] old local id of every kept child's parent)."""
    L = capi.lib()
    coord = np.ascontiguousarray(ch["coord"], dtype=np.float64)
    inp, pinp = capi._sz(np.asarray(ch["inpoel"]).reshape(-1))
    nunk, nie, nn = len(inp) // 4, int(ch["nielem"]), coord.shape[0]
    x, px = capi._f64(coord[:, 0]); y, py = capi._f64(coord[:, 1]); z, pz = capi._f64(coord[:, 2])
    gid, pgid = capi._sz(np.asarray(ch["gid"]))
    ss = ch["sidesets"] or {}
    ids = sorted(ss)
    tri = np.concatenate([np.asarray(ss[s]).reshape(-1, 3) for s in ids]) if ids else np.zeros((0, 3))
    tset = np.concatenate([np.full(len(ss[s]), s, np.int32) for s in ids]) if ids else np.zeros(0, np.int32)
    ntri = len(tset)
    tri, ptri = capi._sz(tri.reshape(-1) if ntri else np.zeros(3))
    tset = np.ascontiguousarray(tset if ntri else np.zeros(1, np.int32), dtype=np.int32)
    nbr = list(ch["nbr_rank"])
    nnbr = len(nbr)
    depth = int(ch.get("depth", 1))
    nb_a = np.ascontiguousarray(nbr or [0], dtype=np.int32)
    rc_a, prc = capi._sz(np.asarray(list(ch["recv_counts"]) or [0]))
    h = C.c_void_p()
    # (two ghost layers: qdg_refine_chunk_depth -- the refined chunk's plan entries are its own)
    capi._chk(L.qdg_refine_chunk_depth(C.c_size_t(nie), C.c_size_t(nunk), C.c_size_t(nn), pinp, px, py, pz, pgid,
                                       C.c_size_t(ntri), ptri, tset.ctypes.data_as(capi.c_i32p), C.c_size_t(nnbr),
                                       nb_a.ctypes.data_as(capi.c_i32p), prc, C.c_int(depth), C.byref(h)))
    try:
        n = [C.c_size_t() for _ in range(5)]
        capi._chk(L.qdg_chunk_refined_sizes(h, *[C.byref(v) for v in n]))
        nie2, nunk2, nn2, ntri2, nsend = (int(v.value) for v in n)
        ne_, ng1_ = C.c_size_t(), C.c_size_t()
        capi._chk(L.qdg_chunk_refined_plan(h, C.byref(ne_), C.byref(ng1_), None, None))
        nnbr, nghost1 = int(ne_.value), int(ng1_.value)
        nb2 = np.zeros(max(1, nnbr), dtype=np.int32); nl2 = np.ones(max(1, nnbr), dtype=np.int32)
        capi._chk(L.qdg_chunk_refined_plan(h, None, None, nb2.ctypes.data_as(capi.c_i32p), nl2.ctypes.data_as(capi.c_i32p)))
        nbr = [int(v) for v in nb2[:nnbr]]
        inp2 = np.empty(4 * nunk2, dtype=np.uint64); gid2 = np.empty(nunk2, dtype=np.uint64)
        par = np.empty(nunk2, dtype=np.uint64)
        c2 = np.empty((3, nn2))
        tri2 = np.zeros(max(1, 3 * ntri2), dtype=np.uint64); tset2 = np.zeros(max(1, ntri2), dtype=np.int32)
        soff = np.zeros(nnbr + 1, dtype=np.uint64); slist = np.zeros(max(1, nsend), dtype=np.uint64)
        rc2 = np.zeros(max(1, nnbr), dtype=np.uint64)
        capi._chk(L.qdg_chunk_refined_get(h, inp2.ctypes.data_as(capi.c_szp), gid2.ctypes.data_as(capi.c_szp),
                                          par.ctypes.data_as(capi.c_szp), c2[0].ctypes.data_as(capi.c_f64p),
                                          c2[1].ctypes.data_as(capi.c_f64p), c2[2].ctypes.data_as(capi.c_f64p),
                                          tri2.ctypes.data_as(capi.c_szp), tset2.ctypes.data_as(capi.c_i32p),
                                          soff.ctypes.data_as(capi.c_szp), slist.ctypes.data_as(capi.c_szp),
                                          rc2.ctypes.data_as(capi.c_szp)))
    finally:
        L.qdg_chunk_refined_destroy(h)
    tri2 = tri2[:3 * ntri2].view(np.int64).reshape(-1, 3); tset2 = tset2[:ntri2]
    soff = soff.view(np.int64); slist = slist.view(np.int64)
    new = {"coord": np.ascontiguousarray(c2.T), "inpoel": inp2.view(np.int64).reshape(-1, 4), "nielem": nie2,
           "sidesets": {int(s_): tri2[tset2 == s_] for s_ in ids if (tset2 == s_).any()},
           "gid": gid2.view(np.int64), "nbr_rank": nbr, "nbr_layer": [int(v) for v in nl2[:nnbr]],
           "depth": depth, "nghost1": nghost1,
           "send_lists": [slist[soff[i]:soff[i + 1]] for i in range(nnbr)],
           "recv_counts": [int(v) for v in rc2[:nnbr]]}
    return new, par.view(np.int64)


def _refine_chunk_numpy(ch):
    """Uniform 1:8 refinement of ONE RANK's chunk of a partitioned mesh (a dict as
    partition.build_chunk / meshgen.kuhn_box_chunk return it), done by the rank alone:
    owned and ghost tets are refined with the same pattern (edge midpoints coincide across the
    chunk boundary, so the refined mesh stays conforming), the children of the owned tets are the
    new owned tets, and the new ghost layer is the set of children of OLD ghosts that share a
    face with a new owned tet.  The halo plan follows without communication: a child of my tet A
    touches a child of rank q's tet B only if A and B touched, i.e. only where both ranks already
    hold the other's tet -- so my send list to q (my children that touch a child of one of q's
    tets) and q's new ghosts from me are the same set, and both sides order it by (global id of
    the parent, child number).  Uniform refinement keeps the load balance of the cut, so this IS
    the re-partition step of BASELINE config 5 (DG::resizePostAMR, DG.cpp:1536-1612 rebuilds
    FaceData and ghosts from the refined chunk the same way).
    Returns (new chunk dict, parent[8*nunk_old] local id of every child's parent)."""
    coord, inpoel, nie = ch["coord"], np.asarray(ch["inpoel"]).reshape(-1, 4), int(ch["nielem"])
    nunk = inpoel.shape[0]
    # side-set triangles that are faces of a local tet (a triangle can have its three nodes in
    # the chunk without being one)
    allf = _rows(_face_keys(inpoel))
    ss_in = {}
    for sid, tri in (ch["sidesets"] or {}).items():
        tri = np.asarray(tri).reshape(-1, 3)
        ok = np.isin(_rows(np.sort(tri, axis=1)), allf)
        if ok.any():
            ss_in[int(sid)] = tri[ok]
    c2, i2, s2, par = refine_uniform(coord, inpoel, ss_in)
    gid = np.asarray(ch["gid"], dtype=np.int64)
    cgid = 8 * gid[par] + np.tile(np.arange(8), nunk)             # global id of a child
    owned = np.arange(8 * nie)                                     # children of owned tets come first
    # faces of the owned children that have no owned partner and are not physical boundary
    fk = _face_keys(i2)
    vo = _rows(fk[:4 * 8 * nie])
    uniq, cnt = np.unique(vo, return_counts=True)
    free = np.isin(vo, uniq[cnt == 1])
    if s2:
        bkey = _rows(np.sort(np.concatenate([np.asarray(t).reshape(-1, 3) for t in s2.values()]), axis=1))
        free &= ~np.isin(vo, bkey)
    # ghost children that own one of those faces; and, the other way round, owned children that
    # share a face with a child of a neighbour's tet
    vg = _rows(fk[4 * 8 * nie:])
    gface_used = np.isin(vg, vo[free])
    ghost_child = np.unique(np.nonzero(gface_used)[0] // 4) + 8 * nie
    oface_used = free & np.isin(vo, vg)
    # owner rank of every old ghost
    off = np.concatenate([[0], np.cumsum(ch["recv_counts"])]).astype(np.int64)
    owner_of_ghost = np.zeros(nunk - nie, dtype=np.int64)
    for i, q in enumerate(ch["nbr_rank"]):
        owner_of_ghost[off[i]:off[i + 1]] = q
    gowner = owner_of_ghost[par[ghost_child] - nie]
    order = np.lexsort((cgid[ghost_child], gowner))                # by owner rank, then global child id
    ghost_child, gowner = ghost_child[order], gowner[order]
    # send lists: owned children by the rank whose tet's child they touch
    send_lists = []
    lookup = {}
    gf = np.nonzero(gface_used)[0]
    for key, g in zip(vg[gf], gf // 4 + 8 * nie):
        lookup[key.tobytes()] = owner_of_ghost[par[g] - nie]
    of = np.nonzero(oface_used)[0]
    dest = np.array([lookup[k.tobytes()] for k in vo[of]], dtype=np.int64) if len(of) else np.zeros(0, np.int64)
    for q in ch["nbr_rank"]:
        mine = np.unique(of[dest == q] // 4)
        send_lists.append(mine[np.argsort(cgid[mine], kind="stable")])
    keep = np.concatenate([owned, ghost_child])
    nodes, inv = np.unique(i2[keep].reshape(-1), return_inverse=True)
    g2l = np.full(c2.shape[0], -1, dtype=np.int64)
    g2l[nodes] = np.arange(len(nodes))
    ss = {}
    for sid, tri in s2.items():
        loc = g2l[np.asarray(tri).reshape(-1, 3)]
        ok = (loc >= 0).all(axis=1)
        if ok.any():
            ss[int(sid)] = loc[ok]
    new = {"coord": c2[nodes], "inpoel": inv.reshape(-1, 4), "nielem": 8 * nie, "sidesets": ss,
           "gid": cgid[keep], "nbr_rank": list(ch["nbr_rank"]),
           "send_lists": send_lists,
           "recv_counts": [int((gowner == q).sum()) for q in ch["nbr_rank"]]}
    return new, par[keep]
