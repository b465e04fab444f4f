"""Host-side mesh-derived data (libqdg mirrors of inciter::FaceData and the
geometry generators) -- no GPU needed.

 * against the reference's own unit-test known answers
   (tests/golden/derived_data_ka.json, extracted from
   tests/unit/Mesh/TestDerivedData.cpp:2429,2767,3096,3160);
 * against the oracle's literal restatement on the regression meshes
   (must be IDENTICAL: same content, same ordering);
 * the synthetic chunk generator: partition covers the mesh, halo plans agree.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_fixture
from oracle import oracle as O
from quinoa_amd import capi, dgmesh, meshgen


@pytest.fixture(scope="module")
def ka():
    with open(os.path.join(GOLDEN, "derived_data_ka.json")) as fh:
        return json.load(fh)


def test_genesuf_known_answer(ka):
    d = ka["genEsuf"]
    inpoel = np.array(d["inpoel_1based"], dtype=np.int64) - 1
    belem = np.array(d["belem_1based"], dtype=np.uint64) - 1
    ne = len(inpoel) // 4
    L = capi.lib()
    esuel = np.zeros(4 * ne, dtype=np.int32)
    inp, pinp = capi._sz(inpoel)
    capi._chk(L.qdg_gen_esuel(ne, pinp, esuel.ctypes.data_as(capi.c_i32p)))
    nipfac = L.qdg_gen_nipfac(ne, d["nbfac"], esuel.ctypes.data_as(capi.c_i32p))
    assert nipfac == d["nipfac"]
    esuf = np.zeros(2 * nipfac, dtype=np.int32)
    capi._chk(L.qdg_gen_esuf(ne, d["nbfac"], belem.ctypes.data_as(capi.c_szp),
                             esuel.ctypes.data_as(capi.c_i32p), esuf.ctypes.data_as(capi.c_i32p)))
    # the reference compares esuf[i] with correct_esuf[i]-1 (TestDerivedData.cpp:2759-2762)
    assert np.array_equal(esuf, np.array(d["correct_esuf_1based"], dtype=np.int32) - 1)
    # oracle restatement gives the same
    om_esup1 = np.zeros(4 * ne + 1, dtype=np.int64); om_esup2 = np.zeros(inpoel.max() + 2, dtype=np.int64)
    OL = O.lib()
    import ctypes as C
    OL.orc_gen_esup(inpoel.ctypes.data_as(O.c_i64p), C.c_int64(ne), C.c_int64(int(inpoel.max()) + 1),
                    om_esup1.ctypes.data_as(O.c_i64p), om_esup2.ctypes.data_as(O.c_i64p))
    oes = np.zeros(4 * ne, dtype=np.int32)
    OL.orc_gen_esuel(inpoel.ctypes.data_as(O.c_i64p), C.c_int64(ne), C.c_int64(int(inpoel.max()) + 1),
                     om_esup1.ctypes.data_as(O.c_i64p), om_esup2.ctypes.data_as(O.c_i64p),
                     oes.ctypes.data_as(O.c_i32p))
    assert np.array_equal(oes, esuel)


def test_geninpofa_known_answer(ka):
    d = ka["genInpofa"]
    inpoel = np.array(d["inpoel_1based"], dtype=np.uint64) - 1
    tri = np.array(d["triinpoel_1based"], dtype=np.uint64) - 1
    ne = len(inpoel) // 4
    L = capi.lib()
    esuel = np.zeros(4 * ne, dtype=np.int32)
    capi._chk(L.qdg_gen_esuel(ne, inpoel.ctypes.data_as(capi.c_szp), esuel.ctypes.data_as(capi.c_i32p)))
    nipfac = L.qdg_gen_nipfac(ne, d["nbfac"], esuel.ctypes.data_as(capi.c_i32p))
    inpofa = np.zeros(3 * nipfac, dtype=np.uint64)
    capi._chk(L.qdg_gen_inpofa(ne, d["nbfac"], inpoel.ctypes.data_as(capi.c_szp),
                               tri.ctypes.data_as(capi.c_szp), esuel.ctypes.data_as(capi.c_i32p),
                               inpofa.ctypes.data_as(capi.c_szp)))
    assert np.array_equal(inpofa.astype(np.int64), np.array(d["correct_inpofa_1based"]) - 1)
    # belem: every boundary face's host element contains its 3 nodes
    belem = np.zeros(d["nbfac"], dtype=np.uint64)
    capi._chk(L.qdg_gen_belem(ne, d["nbfac"], inpoel.ctypes.data_as(capi.c_szp),
                              inpofa.ctypes.data_as(capi.c_szp), belem.ctypes.data_as(capi.c_szp)))
    for f in range(d["nbfac"]):
        assert set(inpofa[3 * f:3 * f + 3]) <= set(inpoel[4 * int(belem[f]):4 * int(belem[f]) + 4])


def test_geoface_geoelem_known_answers(ka):
    eps = np.finfo(float).eps
    d = ka["genGeoFaceTri"]
    coord = np.array(d["coord"]).T.copy()
    g = capi.gen_geoface(4, np.array(d["inpofa"]), coord).reshape(4, 7)
    assert np.allclose(g[:, 0], d["farea"], rtol=0, atol=eps)
    assert np.allclose(g[:, 1:4], np.array(d["fnorm"]).T, rtol=0, atol=eps)
    assert np.allclose(g[:, 4:7], np.array(d["fcent"]).T, rtol=0, atol=eps)
    d = ka["genGeoElemTet"]
    coord = np.array(d["coord"]).T.copy()
    ge = capi.gen_geoelem(np.array(d["inpoel"]), coord)
    assert abs(ge[0] - d["vol"]) <= eps
    assert np.allclose(ge[1:4], d["cent"], rtol=0, atol=eps)


@pytest.mark.parametrize("name", ["sod_dg", "sedov_dgp1", "vortical_flow_dg"])
def test_host_mirror_identical_to_oracle(name):
    fix = load_fixture(name)
    ss = {int(s): fix["ss_tri_%d" % s] for s in fix["ss_ids"]}
    om = O.OracleMesh(fix["coord"], fix["inpoel"], ss)
    c = dgmesh.build_chunk(fix["coord"], fix["inpoel"], None, ss)
    assert np.array_equal(c.triinpoel.astype(np.int64), om.triinpoel)
    for k in om.bface:
        assert np.array_equal(c.bface[k].astype(np.int64), om.bface[k])
    assert np.array_equal(c.esuel, om.esuel)
    assert np.array_equal(c.esuf, om.esuf)
    assert np.array_equal(c.inpofa.astype(np.int64), om.inpofa)
    assert np.array_equal(c.geoFace, om.geoFace)
    assert np.array_equal(c.geoElem, om.geoElem)
    # mesh closure: sum of area*normal over the faces of every element vanishes
    # (the leak test of src/Mesh/DerivedData.cpp:1493-1539)
    s = np.zeros((om.nelem, 3))
    gf = c.geoFace.reshape(-1, 7)
    for f in range(c.nfac):
        el, er = c.esuf[2 * f], c.esuf[2 * f + 1]
        s[el] += gf[f, 0] * gf[f, 1:4]
        if er >= 0:
            s[er] -= gf[f, 0] * gf[f, 1:4]
    assert np.abs(s).max() < 1e-15


def test_non_manifold_mesh_is_an_error_not_a_crash():
    inpoel = np.array([0, 1, 2, 3, 0, 1, 2, 4, 0, 1, 2, 5], dtype=np.uint64)   # 3 tets on one face
    esuel = np.zeros(12, dtype=np.int32)
    rc = capi.lib().qdg_gen_esuel(3, inpoel.ctypes.data_as(capi.c_szp), esuel.ctypes.data_as(capi.c_i32p))
    assert rc != 0 and b"non-manifold" in capi.lib().qdg_last_error()


def test_synthetic_chunks_cover_mesh_and_halo_plans_agree():
    nx, ny, nz = 6, 5, 4
    whole = meshgen.kuhn_box(nx, ny, nz)
    cw = dgmesh.build_chunk(whole["coord"], whole["inpoel"], None, whole["sidesets"])
    assert abs(cw.meshvol - 1.0) < 1e-13 and cw.geoElem[0::4].min() > 0
    assert cw.nbfac == 4 * (nx * ny + ny * nz + nx * nz)
    for parts in ((2, 1, 1), (2, 2, 1), (2, 2, 2)):
        n = parts[0] * parts[1] * parts[2]
        chunks = [meshgen.kuhn_box_chunk(nx, ny, nz, parts=parts, rank=r) for r in range(n)]
        gids = np.concatenate([c["gid"][:c["nielem"]] for c in chunks])
        assert len(gids) == nx * ny * nz * 6 == len(np.unique(gids))
        vol = 0.0
        for r, c in enumerate(chunks):
            off = c["nielem"]
            for q, cnt, sl in zip(c["nbr_rank"], c["recv_counts"], c["send_lists"]):
                o = chunks[q]
                iq = o["nbr_rank"].index(r)
                # my ghosts from q are exactly what q sends me, in the same order
                assert np.array_equal(o["gid"][o["send_lists"][iq]], c["gid"][off:off + cnt])
                assert (sl < c["nielem"]).all()
                off += cnt
            assert off == c["inpoel"].shape[0]
            ck = dgmesh.build_chunk(c["coord"], c["inpoel"], c["nielem"], c["sidesets"])
            # every free face of an owned tet is either physical boundary or has a ghost
            assert (ck.esuel.reshape(-1, 4) == -1).sum() == ck.nbfac
            assert (ck.esuel >= c["nielem"]).sum() == ck.nfac - ck.nipfac
            vol += ck.meshvol
        assert abs(vol - 1.0) < 1e-12


def test_synthetic_mesh_generator_has_no_inverted_tets_and_chunks_agree():
    """meshgen.kuhn_box_chunk: a node whose jitter would give a hex a (nearly) inverted Kuhn tet keeps its
    regular position (found at 220^3: one tet in 6.4e7 inverted at 0.2 h).  The rule is evaluated from global
    ids, so every rank of a decomposition places every node exactly where the single-chunk mesh has it."""
    from quinoa_amd import meshgen
    dims = (6, 5, 4)

    def vols(ch):
        c, t = ch["coord"], ch["inpoel"]
        a, b, d = c[t[:, 1]] - c[t[:, 0]], c[t[:, 2]] - c[t[:, 0]], c[t[:, 3]] - c[t[:, 0]]
        return np.einsum("ij,ij->i", a, np.cross(b, d)) / 6.0

    nominal = 1.0 / (6 * dims[0] * dims[1] * dims[2])
    assert vols(meshgen.kuhn_box(*dims)).min() > 0.02 * nominal
    for jit in (0.2, 0.45):                       # 0.45 h: the rule fires for many hexes
        one = meshgen.kuhn_box_chunk(*dims, parts=(1, 1, 1), rank=0, jitter=jit)
        cen = np.zeros((6 * dims[0] * dims[1] * dims[2], 3))
        cen[one["gid"]] = one["coord"][one["inpoel"]].mean(axis=1)
        for r in range(4):
            ch = meshgen.kuhn_box_chunk(*dims, parts=(2, 2, 1), rank=r, jitter=jit)
            got = ch["coord"][ch["inpoel"]].mean(axis=1)          # owned and ghost tets
            assert np.abs(got - cen[ch["gid"]]).max() < 1e-15
    # the rule changes something at 0.45 h and nothing at 0.2 h on this box
    plain = meshgen._unit_jitter(*np.meshgrid(np.arange(7), np.arange(6), np.arange(5), indexing="ij"), 6, 5, 4, 0.45, 12345)
    assert np.abs(plain).max() > 0.3
    I, J, K = (a.ravel() for a in np.meshgrid(np.arange(7), np.arange(6), np.arange(5), indexing="ij"))
    assert meshgen._near_flat_hex_corner(I, J, K, 6, 5, 4, 0.45, 12345).any()
    assert not meshgen._near_flat_hex_corner(I, J, K, 6, 5, 4, 0.2, 12345).any()
