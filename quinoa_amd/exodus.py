"""ExodusII (netCDF classic, 64-bit offsets) element-field output: ctypes front-end of
qdg_exo_write (quinoa_amd/csrc/qdg_exo.cpp).  Stand-in for tk::ExodusIIMeshWriter as
DG::writeFields drives it (src/Inciter/DG.cpp:1165-1215)."""
import ctypes as C

import numpy as np

from . import capi

# ExodusII side number (1..4) of a TETRA from the set of its local nodes on the side:
# side 1 (1,2,4), 2 (2,3,4), 3 (1,4,3), 4 (1,3,2)
_SIDE = {frozenset((0, 1, 3)): 1, frozenset((1, 2, 3)): 2, frozenset((0, 2, 3)): 3, frozenset((0, 1, 2)): 4}


def sidesets_to_elem_sides(inpoel, sidesets):
    """{id: triangles[n,3]} -> {id: (elem[n], side[n])}: the tet that owns each boundary triangle
    and the ExodusII number of that side"""
    inpoel = np.asarray(inpoel, dtype=np.int64).reshape(-1, 4)
    key = {}
    for e, t in enumerate(inpoel):
        for loc, sd in _SIDE.items():
            key[frozenset(int(t[i]) for i in loc)] = (e, sd)
    out = {}
    for sid, tri in sidesets.items():
        es = [key[frozenset(int(v) for v in t)] for t in np.asarray(tri).reshape(-1, 3)]
        out[int(sid)] = (np.array([a for a, _ in es], dtype=np.uint64), np.array([b for _, b in es], dtype=np.int32))
    return out


def write(path, coord, inpoel, sidesets, names, times, vals, title="quinoa_amd"):
    """vals[time, var, elem]"""
    coord = np.asarray(coord, dtype=np.float64)
    inp, pinp = capi._sz(np.asarray(inpoel).reshape(-1))
    x, px = capi._f64(coord[:, 0]); y, py = capi._f64(coord[:, 1]); z, pz = capi._f64(coord[:, 2])
    es = sidesets_to_elem_sides(inpoel, sidesets or {})
    ids = sorted(es)
    sid, psid = capi._i32(np.array(ids or [0], dtype=np.int32))
    off = np.concatenate([[0], np.cumsum([len(es[s][0]) for s in ids])]).astype(np.uint64)
    off, poff = capi._sz(off)
    el, pel = capi._sz(np.concatenate([es[s][0] for s in ids]) if ids else np.zeros(1, np.uint64))
    sd, psd = capi._i32(np.concatenate([es[s][1] for s in ids]) if ids else np.zeros(1, np.int32))
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    times = np.ascontiguousarray(times, dtype=np.float64)
    assert vals.shape == (len(times), len(names), len(inp) // 4)
    arr = (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
    capi._chk(capi.lib().qdg_exo_write(str(path).encode(), title.encode(), C.c_size_t(coord.shape[0]), px, py, pz,
                                       C.c_size_t(len(inp) // 4), pinp, C.c_size_t(len(ids)), psid, poff, pel, psd,
                                       C.c_size_t(len(names)), arr, C.c_size_t(len(times)),
                                       times.ctypes.data_as(capi.c_f64p), vals.ctypes.data_as(capi.c_f64p)))
