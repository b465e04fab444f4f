"""Synthetic unstructured tetrahedral meshes for the DG CompFlow bench/tests.

A box [0,Lx]x[0,Ly]x[0,Lz] of nx*ny*nz hexahedra, each split into 6 Kuhn
tetrahedra (conforming, all with positive `triple(ba,ca,da)` as
src/Mesh/DerivedData.cpp:1478-1480 requires); interior nodes are jittered by
U(-0.2h, 0.2h) per axis and the local element and node numberings are randomly
permuted to emulate an unstructured input file.

`kuhn_box_chunk` generates ONE partition of a px*py*pz block decomposition
directly (owned tets + the one-layer face-neighbour ghost tets + the halo
plan), so that 8 ranks never have to hold the global 64 M-tet mesh.  All
random quantities are counter-based functions of GLOBAL ids, so every rank
sees the same global mesh.

Side sets: 1 x=0, 2 x=Lx, 3 y=0, 4 y=Ly, 5 z=0, 6 z=Lz.
"""
import itertools

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _uniform(ids, stream):
    """counter-based U[0,1) from global ids (uint64) and a stream number"""
    with np.errstate(over="ignore"):
        h = _splitmix64(ids.astype(np.uint64) * np.uint64(6) + np.uint64(stream))
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


# the 6 Kuhn tets of the unit cube: paths 000 -> e_a -> e_a+e_b -> 111
_KUHN = []
for _perm in itertools.permutations(range(3)):
    v = [np.zeros(3, dtype=np.int64)]
    for a in _perm:
        nv = v[-1].copy()
        nv[a] += 1
        v.append(nv)
    v = np.array(v)
    # orientation: det[e_a, e_a+e_b, e_a+e_b+e_c] = sign(perm); make it positive
    M = (v[1:] - v[0]).astype(float)
    if np.linalg.det(M) < 0:
        v[[2, 3]] = v[[3, 2]]
    _KUHN.append(v)
_KUHN = np.array(_KUHN)            # [6 tets][4 nodes][3 ijk offsets]


def _unit_jitter(i, j, k, nx, ny, nz, jitter, seed):
    """jitter of global node (i, j, k) in units of the cell size, [.., 3]; zero on the boundary"""
    g = (i + (nx + 1) * (j + (ny + 1) * k)).astype(np.uint64) + np.uint64(seed) * np.uint64(1000003)
    inner = (i > 0) & (i < nx) & (j > 0) & (j < ny) & (k > 0) & (k < nz)
    return np.stack([np.where(inner, (2 * _uniform(g, a) - 1) * jitter, 0.0) for a in range(3)], axis=-1)


def _near_flat_hex_corner(ni, nj, nk, nx, ny, nz, jitter, seed, frac=0.02):
    """True for the nodes (global indices ni, nj, nk) that are a corner of a hex with a Kuhn tet whose
    volume, under the plain jitter, is below `frac` of the nominal h^3/6.  Evaluated on the index box of
    the given nodes grown by one hex, from global ids only (cell-size units: a unit hex)."""
    lo = [max(int(v.min()) - 1, 0) for v in (ni, nj, nk)]
    hi = [min(int(v.max()) + 1, n) for v, n in zip((ni, nj, nk), (nx, ny, nz))]      # node index range
    out = np.zeros(len(ni), dtype=bool)
    nhx, nhy = hi[0] - lo[0], hi[1] - lo[1]
    if nhx <= 0 or nhy <= 0 or hi[2] - lo[2] <= 0:
        return out
    bad_nodes = set()
    for k in range(lo[2], hi[2]):                                                       # slab of hexes k
        I, J = np.meshgrid(np.arange(lo[0], hi[0]), np.arange(lo[1], hi[1]), indexing="ij")
        I, J = I.ravel(), J.ravel()
        K = np.full_like(I, k)
        # corner positions of the unit hexes: ijk offset + jitter
        P = {}
        for di, dj, dk in itertools.product((0, 1), repeat=3):
            P[(di, dj, dk)] = np.array([di, dj, dk], dtype=float) + _unit_jitter(I + di, J + dj, K + dk, nx, ny, nz, jitter, seed)
        flat = np.zeros(len(I), dtype=bool)
        for t in _KUHN:
            p0, p1, p2, p3 = (P[tuple(int(x) for x in v)] for v in t)
            vol = np.einsum("ij,ij->i", p1 - p0, np.cross(p2 - p0, p3 - p0))          # 6 x volume, nominal 1
            flat |= vol < frac
        for h in np.nonzero(flat)[0]:
            for di, dj, dk in itertools.product((0, 1), repeat=3):
                bad_nodes.add((int(I[h]) + di, int(J[h]) + dj, k + dk))
    if bad_nodes:
        b = np.array(sorted(bad_nodes), dtype=np.int64)
        key = lambda a, c, d: a + (nx + 1) * (c + (ny + 1) * d)
        out = np.isin(key(ni, nj, nk), key(b[:, 0], b[:, 1], b[:, 2]))
    return out


def _block_ranges(n, p):
    edges = [(n * i) // p for i in range(p + 1)]
    return [(edges[i], edges[i + 1]) for i in range(p)]


def _hex_tets(I, J, K, npx, npy):
    """global node ids [n*6,4] of the Kuhn tets of hexes (I,J,K)"""
    gi = I[:, None, None] + _KUHN[None, :, :, 0]
    gj = J[:, None, None] + _KUHN[None, :, :, 1]
    gk = K[:, None, None] + _KUHN[None, :, :, 2]
    return (gi + npx * (gj + npy * gk)).reshape(-1, 4)


def kuhn_box_chunk(nx, ny, nz, lengths=(1.0, 1.0, 1.0), parts=(1, 1, 1), rank=0,
                   jitter=0.2, seed=12345, shuffle_seed=67890, depth=1):
    """Return one chunk of the box mesh as a dict:
      coord[nnode,3], inpoel[nunk,4] (local node ids; rows [0,nielem) owned,
      rows [nielem,nunk) ghosts grouped by neighbour rank), nielem,
      sidesets {id: triangles[n,3]} (local node ids, owned tets only),
      gid[nunk] global tet ids, nbr_rank[], send_lists[] (local tet ids per
      neighbour, ordered by global id), recv_counts[].
    depth = 2: two ghost layers (qdg_chunk_build_depth's shape: nghost1 layer-1 ghosts, then layer 2; one plan
    entry per (neighbour rank, layer): nbr_rank, nbr_layer, send_lists, recv_counts per entry); the mesh itself
    and the owned tets' order are those of depth 1.
    """
    px, py, pz = parts
    nranks = px * py * pz
    assert 0 <= rank < nranks
    rx, ry, rz = rank % px, (rank // px) % py, rank // (px * py)
    bx, by, bz = _block_ranges(nx, px), _block_ranges(ny, py), _block_ranges(nz, pz)
    (i0, i1), (j0, j1), (k0, k1) = bx[rx], by[ry], bz[rz]
    npx, npy = nx + 1, ny + 1
    lpofa = np.array([[1, 2, 3], [2, 0, 3], [3, 0, 1], [0, 2, 1]])

    # ---- owned tets (random local order emulates an unstructured input) ----
    I, J, K = np.meshgrid(np.arange(i0, i1), np.arange(j0, j1), np.arange(k0, k1), indexing="ij")
    I, J, K = I.ravel(), J.ravel(), K.ravel()
    own_gnode = _hex_tets(I, J, K, npx, npy)
    own_gid = ((I + nx * (J + ny * K))[:, None] * 6 + np.arange(6)[None, :]).reshape(-1)
    surf = np.repeat((I == i0) | (I == i1 - 1) | (J == j0) | (J == j1 - 1) | (K == k0) | (K == k1 - 1), 6)
    nielem = own_gnode.shape[0]
    perm = np.random.default_rng(shuffle_seed + 7919 * rank).permutation(nielem)
    own_gnode, own_gid, surf = own_gnode[perm], own_gid[perm], surf[perm]

    # ---- halo hexes: one layer around the block, clipped to the domain -----
    ei0, ei1 = max(i0 - 1, 0), min(i1 + 1, nx)
    ej0, ej1 = max(j0 - 1, 0), min(j1 + 1, ny)
    ek0, ek1 = max(k0 - 1, 0), min(k1 + 1, nz)
    nbr_rank, send_lists, recv_counts = [], [], []
    nbr_layer, nghost1 = [], 0
    gho_gnode = np.zeros((0, 4), dtype=np.int64)
    gho_gid = np.zeros(0, dtype=np.int64)
    if nranks > 1 and depth == 2:
        # two layers: the tets of the two outer hex layers of the block and of a two-hex ring around it (a face
        # neighbour of a Kuhn tet lies in its own or in a face-adjacent hex), their face adjacency, and the
        # library's ghost-plan builder (qdg_ghost_plan_build) on that -- the rule qdg_chunk_build_depth applies
        from . import capi, partition
        ei0, ei1 = max(i0 - 2, 0), min(i1 + 2, nx)
        ej0, ej1 = max(j0 - 2, 0), min(j1 + 2, ny)
        ek0, ek1 = max(k0 - 2, 0), min(k1 + 2, nz)
        HI, HJ, HK = np.meshgrid(np.arange(ei0, ei1), np.arange(ej0, ej1), np.arange(ek0, ek1), indexing="ij")
        HI, HJ, HK = HI.ravel(), HJ.ravel(), HK.ravel()
        halo = ~((HI >= i0) & (HI < i1) & (HJ >= j0) & (HJ < j1) & (HK >= k0) & (HK < k1))
        HI, HJ, HK = HI[halo], HJ[halo], HK[halo]

        def blk2(v, ranges):
            b = np.zeros_like(v)
            for q, (lo, hi) in enumerate(ranges):
                b[(v >= lo) & (v < hi)] = q
            return b
        h_owner = np.repeat(blk2(HI, bx) + px * (blk2(HJ, by) + py * blk2(HK, bz)), 6)
        h_gnode = _hex_tets(HI, HJ, HK, npx, npy)
        h_gid = ((HI + nx * (HJ + ny * HK))[:, None] * 6 + np.arange(6)[None, :]).reshape(-1)
        own_hex = own_gid // 6
        oi, oj, ok = own_hex % nx, (own_hex // nx) % ny, own_hex // (nx * ny)
        near = (oi < i0 + 2) | (oi >= i1 - 2) | (oj < j0 + 2) | (oj >= j1 - 2) | (ok < k0 + 2) | (ok >= k1 - 2)
        sidx2 = np.nonzero(near)[0]
        c_gnode = np.concatenate([own_gnode[sidx2], h_gnode])
        c_owner = np.concatenate([np.full(len(sidx2), rank, dtype=np.int64), h_owner])
        c_gid = np.concatenate([own_gid[sidx2], h_gid])
        fd_esuel = capi.gen_esuel(c_gnode)
        pl = partition.ghost_plan(fd_esuel, c_owner, c_gid, rank, depth=2)
        gsel = pl["ghost"] - len(sidx2)
        assert (gsel >= 0).all()
        gho_gnode, gho_gid = h_gnode[gsel], h_gid[gsel]
        nbr_rank, nbr_layer, nghost1 = pl["nbr_rank"], pl["nbr_layer"], pl["nghost1"]
        recv_counts = pl["recv_counts"]
        send_lists = [sidx2[sl].astype(np.int64) for sl in pl["send_lists"]]
    elif nranks > 1:
        HI, HJ, HK = np.meshgrid(np.arange(ei0, ei1), np.arange(ej0, ej1), np.arange(ek0, ek1),
                                 indexing="ij")
        HI, HJ, HK = HI.ravel(), HJ.ravel(), HK.ravel()
        halo = ~((HI >= i0) & (HI < i1) & (HJ >= j0) & (HJ < j1) & (HK >= k0) & (HK < k1))
        HI, HJ, HK = HI[halo], HJ[halo], HK[halo]

        def blk(v, ranges):
            b = np.zeros_like(v)
            for q, (lo, hi) in enumerate(ranges):
                b[(v >= lo) & (v < hi)] = q
            return b
        h_owner = np.repeat(blk(HI, bx) + px * (blk(HJ, by) + py * blk(HK, bz)), 6)
        h_gnode = _hex_tets(HI, HJ, HK, npx, npy)
        h_gid = ((HI + nx * (HJ + ny * HK))[:, None] * 6 + np.arange(6)[None, :]).reshape(-1)
        # faces of surface-layer owned tets vs faces of halo tets (sort-based)
        sidx = np.nonzero(surf)[0]
        fa = np.sort(own_gnode[sidx][:, lpofa].reshape(-1, 3), axis=1)
        fb = np.sort(h_gnode[:, lpofa].reshape(-1, 3), axis=1)
        nn_glob = (nx + 1) * (ny + 1) * (nz + 1)
        ka = (fa[:, 0] * nn_glob + fa[:, 1]) * nn_glob + fa[:, 2] if nn_glob < 2_000_000 else None
        if ka is not None:
            kb = (fb[:, 0] * nn_glob + fb[:, 1]) * nn_glob + fb[:, 2]
        else:   # avoid int64 overflow on very large meshes: structured keys
            ka = np.ascontiguousarray(fa).view([("", fa.dtype)] * 3).reshape(-1)
            kb = np.ascontiguousarray(fb).view([("", fb.dtype)] * 3).reshape(-1)
        ob = np.argsort(kb, kind="stable")
        pos = np.searchsorted(kb[ob], ka)
        pos = np.minimum(pos, len(ob) - 1)
        hit = kb[ob][pos] == ka
        pair_own = sidx[np.nonzero(hit)[0] // 4]          # local owned tet ids
        pair_gho = ob[pos[hit]] // 4                       # halo tet ids
        gsel = np.unique(pair_gho)
        order = np.lexsort((h_gid[gsel], h_owner[gsel]))
        gsel = gsel[order]
        gho_gnode, gho_gid = h_gnode[gsel], h_gid[gsel]
        nbr_rank = [int(q) for q in np.unique(h_owner[gsel])]
        for q in nbr_rank:
            recv_counts.append(int(np.sum(h_owner[gsel] == q)))
            cand = np.unique(pair_own[h_owner[pair_gho] == q])
            send_lists.append(cand[np.argsort(own_gid[cand])].astype(np.int64))
        nbr_layer, nghost1 = [1] * len(nbr_rank), len(gho_gid)

    # ---- local nodes: the block's node box + extra ghost nodes, permuted ----
    bnx, bny, bnz = i1 - i0 + 1, j1 - j0 + 1, k1 - k0 + 1
    nbox = bnx * bny * bnz

    def box_local(g):
        gi, gj, gk = g % npx, (g // npx) % npy, g // (npx * npy)
        inside = (gi >= i0) & (gi <= i1) & (gj >= j0) & (gj <= j1) & (gk >= k0) & (gk <= k1)
        return (gi - i0) + bnx * ((gj - j0) + bny * (gk - k0)), inside
    own_loc, _ = box_local(own_gnode)
    bi, bj, bk = np.meshgrid(np.arange(i0, i1 + 1), np.arange(j0, j1 + 1), np.arange(k0, k1 + 1),
                             indexing="ij")
    # box_local order: i fastest
    g_box = (bi + npx * (bj + npy * bk)).transpose(2, 1, 0).reshape(-1)
    g_used = g_box
    gho_loc = np.zeros((0, 4), dtype=np.int64)
    if len(gho_gnode):
        gl, inside = box_local(gho_gnode)
        extra, inv = np.unique(gho_gnode[~inside], return_inverse=True)
        gho_loc = gl.copy()
        gho_loc[~inside] = nbox + inv
        g_used = np.concatenate([g_box, extra])
    nn = len(g_used)
    nperm = np.random.default_rng(shuffle_seed + 104729 * rank + 1).permutation(nn)
    inpoel = nperm[np.concatenate([own_loc, gho_loc])]
    ni = g_used % npx
    nj = (g_used // npx) % npy
    nk = g_used // (npx * npy)
    hx, hy, hz = lengths[0] / nx, lengths[1] / ny, lengths[2] / nz
    cx, cy, cz = ni * hx, nj * hy, nk * hz
    interior = (ni > 0) & (ni < nx) & (nj > 0) & (nj < ny) & (nk > 0) & (nk < nz)
    if jitter > 0:
        gid64 = g_used.astype(np.uint64) + np.uint64(seed) * np.uint64(1000003)
        # a node keeps its jitter unless one of the (up to 8) hexes around it would get a Kuhn tet of
        # less than 2 % of the nominal volume -- one tet in 6.4e7 at 0.2 h comes out inverted (found at
        # 220^3), and DerivedData.cpp:1478-1480 requires positive volumes.  The rule is a function of
        # global ids alone (the jitter of any node can be evaluated anywhere), so all ranks agree.
        keep = interior & ~_near_flat_hex_corner(ni, nj, nk, nx, ny, nz, jitter, seed)
        cx = cx + np.where(keep, (2 * _uniform(gid64, 0) - 1) * jitter * hx, 0.0)
        cy = cy + np.where(keep, (2 * _uniform(gid64, 1) - 1) * jitter * hy, 0.0)
        cz = cz + np.where(keep, (2 * _uniform(gid64, 2) - 1) * jitter * hz, 0.0)
    coord = np.zeros((nn, 3))
    coord[nperm, 0], coord[nperm, 1], coord[nperm, 2] = cx, cy, cz

    # ---- side sets: faces of owned surface-layer tets on a domain plane ----
    sidx = np.nonzero(surf)[0]
    fo = own_gnode[sidx][:, lpofa]                            # [n][4][3] global nodes
    fi, fj, fk = fo % npx, (fo // npx) % npy, fo // (npx * npy)
    lfo = inpoel[sidx][:, lpofa]
    sidesets = {}
    for sid, arr, val in ((1, fi, 0), (2, fi, nx), (3, fj, 0), (4, fj, ny), (5, fk, 0), (6, fk, nz)):
        on = np.all(arr == val, axis=2)
        sidesets[sid] = lfo[on].reshape(-1, 3).astype(np.int64)

    return {"coord": coord, "inpoel": inpoel.astype(np.int64), "nielem": nielem,
            "sidesets": sidesets, "gid": np.concatenate([own_gid, gho_gid]).astype(np.int64),
            "nbr_rank": nbr_rank, "send_lists": send_lists, "recv_counts": recv_counts,
            "nbr_layer": nbr_layer, "nghost1": int(nghost1), "depth": int(depth) if nranks > 1 else 1,
            "ntet_global": nx * ny * nz * 6}


def kuhn_box(nx, ny, nz, lengths=(1.0, 1.0, 1.0), **kw):
    """Whole box as a single chunk (no ghosts)."""
    return kuhn_box_chunk(nx, ny, nz, lengths, (1, 1, 1), 0, **kw)


def parts_for(nranks):
    """Block decomposition used by bench.py for N ranks of one node."""
    return {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(nranks) or \
        (nranks, 1, 1)
