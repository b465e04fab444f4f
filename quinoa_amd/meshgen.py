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


def _block_ranges(n, p):
    edges = [(n * i) // p for i in range(p + 1)]
    return [(edges[i], edges[i + 1]) for i in range(p)]


def kuhn_box_chunk(nx, ny, nz, lengths=(1.0, 1.0, 1.0), parts=(1, 1, 1), rank=0,
                   jitter=0.2, seed=12345, shuffle_seed=67890):
    """Return one chunk of the box mesh as a dict:
      coord[nnode,3], inpoel[nunk,4] (local node ids; rows [0,nielem) owned,
      rows [nielem,nunk) ghosts grouped by neighbour rank), nielem,
      sidesets {id: triangles[n,3]} (local node ids, owned tets only),
      gid[nunk] global tet ids, nbr_rank[], send_lists[] (local tet ids per
      neighbour, ordered by global id), recv_counts[].
    """
    px, py, pz = parts
    nranks = px * py * pz
    assert 0 <= rank < nranks
    rx, ry, rz = rank % px, (rank // px) % py, rank // (px * py)
    (i0, i1), (j0, j1), (k0, k1) = _block_ranges(nx, px)[rx], _block_ranges(ny, py)[ry], \
        _block_ranges(nz, pz)[rz]
    # extended hex range: one layer around the owned block, clipped
    ei0, ei1 = max(i0 - 1, 0), min(i1 + 1, nx)
    ej0, ej1 = max(j0 - 1, 0), min(j1 + 1, ny)
    ek0, ek1 = max(k0 - 1, 0), min(k1 + 1, nz)
    I, J, K = np.meshgrid(np.arange(ei0, ei1), np.arange(ej0, ej1), np.arange(ek0, ek1),
                          indexing="ij")
    I, J, K = I.ravel(), J.ravel(), K.ravel()
    owned_hex = (I >= i0) & (I < i1) & (J >= j0) & (J < j1) & (K >= k0) & (K < k1)

    def owner_of(i, j, k):
        def blk(v, n, p):
            # inverse of _block_ranges: largest b with (n*b)//p <= v
            b = np.minimum((v * p + p - 1) // n, p - 1)
            lo = (n * b) // p
            b = np.where(lo > v, b - 1, b)
            hi = (n * (b + 1)) // p
            b = np.where(hi <= v, b + 1, b)
            return b
        return blk(i, nx, px) + px * (blk(j, ny, py) + py * blk(k, nz, pz))

    hex_owner = owner_of(I, J, K)
    hex_gid = I + nx * (J + ny * K)
    nh = len(I)
    # tets of all extended hexes: global node ids
    npx, npy = nx + 1, ny + 1
    off = _KUHN                                             # [6][4][3]
    gi = I[:, None, None] + off[None, :, :, 0]
    gj = J[:, None, None] + off[None, :, :, 1]
    gk = K[:, None, None] + off[None, :, :, 2]
    gnode = (gi + npx * (gj + npy * gk)).reshape(nh * 6, 4)
    tet_gid = (hex_gid[:, None] * 6 + np.arange(6)[None, :]).reshape(-1)
    tet_owner = np.repeat(hex_owner, 6)
    tet_owned = np.repeat(owned_hex, 6)

    # face adjacency on the extended set (sort-based)
    lpofa = np.array([[1, 2, 3], [2, 0, 3], [3, 0, 1], [0, 2, 1]])
    nt = gnode.shape[0]
    fn = np.sort(gnode[:, lpofa].reshape(nt * 4, 3), axis=1)
    order = np.lexsort((fn[:, 2], fn[:, 1], fn[:, 0]))
    fs = fn[order]
    same = np.all(fs[1:] == fs[:-1], axis=1)
    a = order[:-1][same] // 4
    b = order[1:][same] // 4
    # ghosts: non-owned tets sharing a face with an owned tet
    ghost_mask = np.zeros(nt, dtype=bool)
    ga = tet_owned[a] & ~tet_owned[b]
    gb = tet_owned[b] & ~tet_owned[a]
    ghost_mask[b[ga]] = True
    ghost_mask[a[gb]] = True
    # send candidates: owned tets adjacent to a tet owned by rank q
    pair_own = np.concatenate([a[ga], b[gb]])
    pair_gho = np.concatenate([b[ga], a[gb]])

    own_idx = np.nonzero(tet_owned)[0]
    # emulate an unstructured input: random local order of the owned tets
    rng = np.random.default_rng(shuffle_seed + 7919 * rank)
    own_idx = own_idx[rng.permutation(len(own_idx))]
    gho_idx = np.nonzero(ghost_mask)[0]
    # ghosts grouped by owner rank, ascending global id within a rank
    gorder = np.lexsort((tet_gid[gho_idx], tet_owner[gho_idx]))
    gho_idx = gho_idx[gorder]
    nbr_rank = np.unique(tet_owner[gho_idx]).astype(np.int64)
    recv_counts = [int(np.sum(tet_owner[gho_idx] == q)) for q in nbr_rank]

    sel = np.concatenate([own_idx, gho_idx])
    nielem, nunk = len(own_idx), len(sel)
    loc_of = np.full(nt, -1, dtype=np.int64)
    loc_of[sel] = np.arange(nunk)

    send_lists = []
    for q in nbr_rank:
        m = tet_owner[pair_gho] == q
        cand = np.unique(pair_own[m])
        cand = cand[np.argsort(tet_gid[cand])]
        send_lists.append(loc_of[cand])

    # local nodes: unique global ids, randomly permuted
    g_used, inv = np.unique(gnode[sel].reshape(-1), return_inverse=True)
    nn = len(g_used)
    nperm = np.random.default_rng(shuffle_seed + 104729 * rank + 1).permutation(nn)
    inpoel = nperm[inv].reshape(nunk, 4)
    ni = g_used % npx
    nj = (g_used // npx) % npy
    nk = g_used // (npx * npy)
    hx, hy, hz = lengths[0] / nx, lengths[1] / ny, lengths[2] / nz
    cx, cy, cz = ni * hx, nj * hy, nk * hz
    interior = (ni > 0) & (ni < nx) & (nj > 0) & (nj < ny) & (nk > 0) & (nk < nz)
    if jitter > 0:
        gid64 = g_used.astype(np.uint64) + np.uint64(seed) * np.uint64(1000003)
        cx = cx + np.where(interior, (2 * _uniform(gid64, 0) - 1) * jitter * hx, 0.0)
        cy = cy + np.where(interior, (2 * _uniform(gid64, 1) - 1) * jitter * hy, 0.0)
        cz = cz + np.where(interior, (2 * _uniform(gid64, 2) - 1) * jitter * hz, 0.0)
    coord = np.zeros((nn, 3))
    coord[nperm, 0], coord[nperm, 1], coord[nperm, 2] = cx, cy, cz

    # side sets from the owned tets' faces lying on a domain plane
    own_nodes = gnode[own_idx]
    fo = own_nodes[:, lpofa]                                   # [n][4][3] global nodes
    fi, fj, fk = fo % npx, (fo // npx) % npy, fo // (npx * npy)
    sidesets = {}
    lfo = inpoel[:nielem][:, lpofa]
    for sid, arr, val in ((1, fi, 0), (2, fi, nx), (3, fj, 0), (4, fj, ny), (5, fk, 0), (6, fk, nz)):
        on = np.all(arr == val, axis=2)
        sidesets[sid] = lfo[on].reshape(-1, 3).astype(np.int64)

    return {"coord": coord, "inpoel": inpoel.astype(np.int64), "nielem": nielem,
            "sidesets": sidesets, "gid": tet_gid[sel].astype(np.int64),
            "nbr_rank": [int(q) for q in nbr_rank], "send_lists": send_lists,
            "recv_counts": recv_counts, "ntet_global": nx * ny * nz * 6}


def kuhn_box(nx, ny, nz, lengths=(1.0, 1.0, 1.0), **kw):
    """Whole box as a single chunk (no ghosts)."""
    return kuhn_box_chunk(nx, ny, nz, lengths, (1, 1, 1), 0, **kw)


def parts_for(nranks):
    """Block decomposition used by bench.py for N ranks of one node."""
    return {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(nranks) or \
        (nranks, 1, 1)
