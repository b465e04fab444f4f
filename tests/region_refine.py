"""Test tooling: a conforming REGION refinement of a tetrahedron mesh -- marked tets 1:8, closure
with the 1:2 and 1:4 templates, everything else promoted to 1:8 until the mesh is conforming
(the red-green rule set of the reference's AMR library, src/Inciter/AMR/refinement.hpp:78-424,
mesh_adapter.cpp).  It stands in for the reference's Refiner as the PRODUCER of a refined
connectivity + parent-per-tet that a caller hands qdg_mesh_from_connectivity /
qdg_state_transfer; the library itself accepts any such mesh.  Plain Python: small meshes only."""
import numpy as np

EDGES = [(0, 1), (1, 2), (0, 2), (0, 3), (1, 3), (2, 3)]
FACES = [(1, 2, 3), (2, 0, 3), (3, 0, 1), (0, 2, 1)]       # lpofa: face opposite local node f... by index


def refine_region(coord, inpoel, sidesets, marked):
    """returns (coord2, inpoel2, sidesets2, parent) -- children of tet e are consecutive"""
    coord = np.asarray(coord, dtype=np.float64)
    inpoel = np.asarray(inpoel, dtype=np.int64).reshape(-1, 4)
    ne = inpoel.shape[0]
    red = np.asarray(marked, dtype=bool).copy()
    ekey = lambda a, b: (a, b) if a < b else (b, a)
    tet_edges = [[ekey(int(t[i]), int(t[j])) for i, j in EDGES] for t in inpoel]
    while True:
        medges = set()
        for e in np.nonzero(red)[0]:
            medges.update(tet_edges[e])
        changed = False
        kind = np.zeros(ne, dtype=np.int64)              # 0 keep, 2, 4, 8
        info = [None] * ne
        for e in range(ne):
            if red[e]:
                kind[e] = 8
                continue
            m = [k for k, ed in enumerate(tet_edges[e]) if ed in medges]
            if not m:
                continue
            if len(m) == 1:
                kind[e] = 2; info[e] = EDGES[m[0]]
                continue
            if len(m) == 3:
                nodes = set()
                for k in m:
                    nodes.update(EDGES[k])
                if len(nodes) == 3:                       # the three edges of one face
                    kind[e] = 4; info[e] = tuple(sorted(nodes))
                    continue
            red[e] = True; changed = True
        if not changed:
            break
    mid = {}
    newc = []

    def m_(a, b):
        k = ekey(int(a), int(b))
        if k not in mid:
            mid[k] = coord.shape[0] + len(newc)
            newc.append(0.5 * (coord[k[0]] + coord[k[1]]))
        return mid[k]

    out, parent = [], []

    def emit(e, t):
        out.append(t); parent.append(e)

    for e in range(ne):
        t = [int(v) for v in inpoel[e]]
        if kind[e] == 0:
            emit(e, t)
        elif kind[e] == 2:
            i, j = info[e]
            m = m_(t[i], t[j])
            a = list(t); a[j] = m; emit(e, a)
            b = list(t); b[i] = m; emit(e, b)
        elif kind[e] == 4:
            p, q, r = info[e]
            mpq, mqr, mrp = m_(t[p], t[q]), m_(t[q], t[r]), m_(t[r], t[p])
            a = list(t); a[q] = mpq; a[r] = mrp; emit(e, a)
            a = list(t); a[p] = mpq; a[r] = mqr; emit(e, a)
            a = list(t); a[p] = mrp; a[q] = mqr; emit(e, a)
            a = list(t); a[p] = mpq; a[q] = mqr; a[r] = mrp; emit(e, a)
        else:
            mm = {(i, j): m_(t[i], t[j]) for i, j in EDGES}
            g = lambda i, j: mm[(i, j)] if (i, j) in mm else mm[(j, i)]
            for v in range(4):                            # corner children
                a = [g(v, w) if w != v else t[v] for w in range(4)]
                emit(e, a)
            A, B = g(0, 2), g(1, 3)                       # inner octahedron, split along m02-m13
            ring = [g(0, 1), g(1, 2), g(2, 3), g(0, 3)]
            for k in range(4):
                emit(e, [A, B, ring[k], ring[(k + 1) % 4]])
    c2 = np.vstack([coord, np.array(newc).reshape(-1, 3)]) if newc else coord.copy()
    i2 = np.array(out, dtype=np.int64)
    # orientation: positive volume for every child
    a, b, d = c2[i2[:, 1]] - c2[i2[:, 0]], c2[i2[:, 2]] - c2[i2[:, 0]], c2[i2[:, 3]] - c2[i2[:, 0]]
    vol = np.einsum("ij,ij->i", a, np.cross(b, d))
    neg = vol < 0
    i2[neg, 2], i2[neg, 3] = i2[neg, 3].copy(), i2[neg, 2].copy()
    ss2 = {}
    for sid, tri in (sidesets or {}).items():
        res = []
        for tr in np.asarray(tri, dtype=np.int64).reshape(-1, 3):
            p, q, r = (int(v) for v in tr)
            me = [ekey(p, q) in mid, ekey(q, r) in mid, ekey(r, p) in mid]
            if sum(me) == 0:
                res.append([p, q, r])
            elif sum(me) == 3:
                mpq, mqr, mrp = mid[ekey(p, q)], mid[ekey(q, r)], mid[ekey(r, p)]
                res += [[p, mpq, mrp], [mpq, q, mqr], [mrp, mqr, r], [mpq, mqr, mrp]]
            elif sum(me) == 1:
                if me[0]:
                    m = mid[ekey(p, q)]; res += [[p, m, r], [m, q, r]]
                elif me[1]:
                    m = mid[ekey(q, r)]; res += [[p, q, m], [p, m, r]]
                else:
                    m = mid[ekey(r, p)]; res += [[p, q, m], [m, q, r]]
            else:
                raise AssertionError("non-conforming boundary triangle")
        ss2[int(sid)] = np.array(res, dtype=np.int64)
    return c2, i2, ss2, np.array(parent, dtype=np.int64)


def check_conforming(inpoel, sidesets):
    """every face is shared by exactly two tets or is a boundary triangle (each once)"""
    inpoel = np.asarray(inpoel, dtype=np.int64).reshape(-1, 4)
    f = np.sort(inpoel[:, FACES].reshape(-1, 3), axis=1)
    keys, cnt = np.unique(f, axis=0, return_counts=True)
    assert cnt.max() <= 2
    free = {tuple(k) for k in keys[cnt == 1]}
    bnd = [tuple(sorted(int(v) for v in t)) for tri in sidesets.values() for t in np.asarray(tri).reshape(-1, 3)]
    assert len(bnd) == len(set(bnd))
    assert free == set(bnd), (len(free), len(bnd))
