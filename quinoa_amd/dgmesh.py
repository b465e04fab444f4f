"""Host-side assembly of one DG mesh chunk in the reference's data model.

What Inciter's `DG` chare holds after its setup phase (src/Inciter/DG.cpp:46-103
constructor + the ghost set-up of :134-949), produced here from a chunk
description (owned tets, ghost tets, side-set triangles):

  * `FaceData` of the OWNED tets (esuel, inpofa, esuf, belem, bface, triinpoel)
    via libqdg's mirrors of src/Inciter/FaceData.cpp:19-41;
  * chare-boundary faces appended after the `nipfac` interior/physical faces
    with the ghost as right element (DG.cpp:363,480-483), `esuel` of the owned
    tets pointing at ghost ids >= nielem (DG::addEsuel, DG.cpp:810-856), face
    nodes ordered so that the normal points out of the owned tet
    (DG.cpp:661-663);
  * geoFace for all faces, geoElem for owned + ghost tets.

Charm++ messaging, over-decomposition and AMR stay out of scope; this module is
the serial stand-in for that set-up so the hot path can be driven and tested.
"""
import numpy as np

from . import capi

LPOFA = np.array([[1, 2, 3], [2, 0, 3], [3, 0, 1], [0, 2, 1]])


class Chunk:
    pass


def build_chunk(coord, inpoel, nielem=None, sidesets=None):
    coord = np.ascontiguousarray(coord, dtype=np.float64)
    inpoel = np.ascontiguousarray(inpoel, dtype=np.int64).reshape(-1, 4)
    nunk = inpoel.shape[0]
    nielem = nunk if nielem is None else int(nielem)
    c = Chunk()
    c.coord, c.inpoel, c.nielem, c.nunk = coord, inpoel, nielem, nunk
    own = inpoel[:nielem]
    c.bface, c.triinpoel = capi.bnd_faces(own, sidesets or {})
    fd = capi.FaceData(own, c.bface, c.triinpoel)
    c.nbfac, c.nipfac = fd.nbfac, fd.nipfac
    esuel = fd.esuel.copy()
    esuf = fd.esuf
    inpofa = fd.inpofa.astype(np.int64)
    if nunk > nielem:
        # free faces of owned tets: no neighbour and not a physical-boundary face
        free = np.nonzero(esuel.reshape(-1, 4) == -1)
        fe, flf = free[0], free[1]
        fnodes = own[fe[:, None], LPOFA[flf]]                      # [n,3] outward order
        isb = np.zeros(len(fe), dtype=bool)
        if c.nbfac:
            bkey = np.sort(c.triinpoel.astype(np.int64), axis=1)
            isb = _isin_rows(np.sort(fnodes, axis=1), bkey)
        fe, flf, fnodes = fe[~isb], flf[~isb], fnodes[~isb]
        # ghost faces
        gh = inpoel[nielem:]
        gnodes = gh[:, LPOFA].reshape(-1, 3)
        gid = np.repeat(np.arange(nielem, nunk), 4)
        ka = np.sort(fnodes, axis=1)
        kb = np.sort(gnodes, axis=1)
        ia, ib = _match_rows(ka, kb)
        if len(ia) != len(fe) and len(np.unique(gid[ib])) != nunk - nielem:
            raise capi.QdgError("build_chunk: ghost layer does not match the chunk's free faces")
        esuel[4 * fe[ia] + flf[ia]] = gid[ib]
        esuf = np.concatenate([esuf, np.stack([fe[ia], gid[ib]], axis=1).reshape(-1).astype(np.int32)])
        inpofa = np.concatenate([inpofa, fnodes[ia].reshape(-1)])
    c.esuel = np.ascontiguousarray(esuel, dtype=np.int32)
    c.esuf = np.ascontiguousarray(esuf, dtype=np.int32)
    c.inpofa = np.ascontiguousarray(inpofa, dtype=np.uint64)
    c.nfac = len(c.esuf) // 2
    c.geoFace = capi.gen_geoface(c.nfac, c.inpofa, coord)
    c.geoElem = capi.gen_geoelem(inpoel, coord)
    c.meshvol = float(c.geoElem[0:4 * nielem:4].sum())
    return c


def _row_view(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a.view([("", a.dtype)] * a.shape[1]).reshape(-1)


def _isin_rows(a, b):
    return np.isin(_row_view(a), _row_view(b))


def _match_rows(a, b):
    """indices (ia, ib) with a[ia] == b[ib] row-wise (each row at most once)"""
    va, vb = _row_view(a), _row_view(b)
    oa, ob = np.argsort(va, kind="stable"), np.argsort(vb, kind="stable")
    sa, sb = va[oa], vb[ob]
    pos = np.searchsorted(sb, sa)
    pos = np.minimum(pos, len(sb) - 1) if len(sb) else pos
    ok = (sb[pos] == sa) if len(sb) else np.zeros(len(sa), dtype=bool)
    return oa[ok], ob[pos[ok]]


def upload(ctx, chunk, elem_gid=None):
    """qdg_mesh_upload[_gid] of a chunk (capi.Mesh); elem_gid: global tet ids, faces oriented by them."""
    return capi.Mesh(ctx, chunk.nielem, chunk.inpoel, chunk.coord, chunk.esuf, chunk.esuel,
                     chunk.inpofa, chunk.geoFace, chunk.geoElem, chunk.bface, chunk.nbfac, elem_gid=elem_gid)
