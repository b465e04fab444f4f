"""Mesh-derived data generated on the GPU (qdg_dev_facedata: sort/scan kernels)
against the host mirror of inciter::FaceData / DerivedData -- integer arrays bit
for bit, geometry to rounding -- and against the reference's known answers."""
import json
import os
import time

import numpy as np
import pytest

from conftest import ROOT, load_fixture

pytestmark = pytest.mark.gpu


def _compare(ctx, inpoel, coord, sidesets):
    from quinoa_amd import capi
    bface, tri = capi.bnd_faces(inpoel, sidesets)
    fd = capi.FaceData(inpoel, bface, tri)
    g = capi.dev_facedata(ctx, inpoel, coord, tri)
    assert g["nipfac"] == fd.nipfac
    assert np.array_equal(g["esuel"], fd.esuel)
    assert np.array_equal(g["inpofa"], fd.inpofa.astype(np.uint64))
    assert np.array_equal(g["esuf"], fd.esuf)
    assert np.array_equal(g["belem"], np.asarray(fd.belem, dtype=np.uint64))
    gf = capi.gen_geoface(fd.nipfac, fd.inpofa, coord)
    ge = capi.gen_geoelem(inpoel, coord)
    assert np.abs(g["geoFace"] - gf).max() <= 1e-14 * max(1.0, np.abs(gf).max())
    assert np.abs(g["geoElem"] - ge).max() <= 1e-14 * max(1.0, np.abs(ge).max())
    return g


def test_device_facedata_equals_host_on_reference_meshes(cases):
    from quinoa_amd import capi
    ctx = capi.Context(1, cfl=0.3)
    try:
        for name in ("sod_dg", "sedov_dgp1", "taylor_green_dgp2", "slot_cyl_dg"):
            fix = load_fixture(name)
            ss = {int(s): fix["ss_tri_%d" % s] for s in fix["ss_ids"]}
            _compare(ctx, fix["inpoel"], fix["coord"], ss)
    finally:
        ctx.close()


def test_device_facedata_known_answers_and_tiny_meshes():
    """the reference's unit-test mesh (tests/unit/Mesh/TestDerivedData.cpp) and 1-/6-tet meshes"""
    from quinoa_amd import capi, meshgen
    ka = json.load(open(os.path.join(ROOT, "tests", "golden", "derived_data_ka.json")))
    ctx = capi.Context(1, cfl=0.3)
    try:
        # TestDerivedData.cpp:2767 (genInpofa) and :2429 (genEsuf): mesh + boundary triangles
        d = ka["genInpofa"]
        inpoel = (np.array(d["inpoel_1based"], dtype=np.int64) - 1).reshape(-1, 4)
        tri = (np.array(d["triinpoel_1based"], dtype=np.int64) - 1).reshape(-1, 3)
        nnode = int(inpoel.max()) + 1
        coord = np.random.default_rng(1).normal(size=(nnode, 3))
        g = capi.dev_facedata(ctx, inpoel, coord, tri)
        assert np.array_equal(g["inpofa"].astype(np.int64), np.array(d["correct_inpofa_1based"]) - 1)
        d = ka["genEsuf"]
        inpoel2 = (np.array(d["inpoel_1based"], dtype=np.int64) - 1).reshape(-1, 4)
        if np.array_equal(inpoel2, inpoel) and d["nbfac"] == len(tri):
            assert g["nipfac"] == d["nipfac"]
            assert np.array_equal(g["esuf"], np.array(d["correct_esuf_1based"], dtype=np.int32) - 1)
        one = np.array([[0, 1, 2, 3]])
        c1 = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=float)
        g = capi.dev_facedata(ctx, one, c1, np.array([[1, 2, 3], [2, 0, 3], [3, 0, 1], [0, 2, 1]]))
        assert g["nipfac"] == 4 and (g["esuel"] == -1).all() and (g["esuf"][1::2] == -1).all()
        assert abs(g["geoElem"][0] - 1.0 / 6.0) < 1e-15
        ch = meshgen.kuhn_box(1, 1, 1)
        _compare(ctx, ch["inpoel"], ch["coord"], ch["sidesets"])
    finally:
        ctx.close()


def test_inverted_tet_does_not_hide_an_unmatched_boundary_face():
    """qdg_dev_facedata tolerates inverted tets (the reference's derived-data unit meshes are not all
    positively oriented) but must still refuse a boundary triangle that is a face of no tet -- the two
    conditions have separate error bits on the device (round 3's build lost the second behind the first)."""
    from quinoa_amd import capi, meshgen
    ch = meshgen.kuhn_box(2, 2, 2)
    inp = ch["inpoel"].copy()
    inp[5, [0, 1]] = inp[5, [1, 0]]                     # one inverted tet
    _, tri = capi.bnd_faces(inp, ch["sidesets"])
    ctx = capi.Context(1, cfl=0.3)
    try:
        g = capi.dev_facedata(ctx, inp, ch["coord"], tri)      # inverted tet alone: accepted here
        assert g["geoElem"][4 * 5] < 0.0
        # a triangle of three nodes that no tet has as a face
        n = ch["coord"].shape[0]
        bogus = np.array([[0, n // 2, n - 1]], dtype=np.uint64)
        faces = {tuple(sorted(inp[e][list(f)])) for e in range(len(inp)) for f in ([1, 2, 3], [2, 0, 3], [3, 0, 1], [0, 2, 1])}
        assert tuple(sorted(int(v) for v in bogus[0])) not in faces
        with pytest.raises(capi.QdgError, match="not a face of any tet"):
            capi.dev_facedata(ctx, inp, ch["coord"], np.concatenate([tri.astype(np.uint64), bogus]))
        with pytest.raises(capi.QdgError, match="non-positive element volume"):
            capi.mesh_from_connectivity(ctx, inp, ch["coord"], ch["sidesets"])
    finally:
        ctx.close()


def test_device_facedata_full_size_and_timing():
    """998 250 tets: identical to the host mirror; the timing is printed for DESIGN.md"""
    from quinoa_amd import capi, meshgen
    ch = meshgen.kuhn_box(55, 55, 55)
    ctx = capi.Context(1, cfl=0.3)
    try:
        t0 = time.perf_counter()
        bface, tri = capi.bnd_faces(ch["inpoel"], ch["sidesets"])
        fd = capi.FaceData(ch["inpoel"], bface, tri)
        gf = capi.gen_geoface(fd.nipfac, fd.inpofa, ch["coord"])
        ge = capi.gen_geoelem(ch["inpoel"], ch["coord"])
        t1 = time.perf_counter()
        capi.dev_facedata(ctx, ch["inpoel"], ch["coord"], tri)      # warm-up (module load)
        t2 = time.perf_counter()
        g = capi.dev_facedata(ctx, ch["inpoel"], ch["coord"], tri)
        t3 = time.perf_counter()
        print("\nFaceData + geometry of 998250 tets: host %.2f s (incl. bnd_faces), device %.3f s "
              "(incl. PCIe both ways)" % (t1 - t0, t3 - t2))
        assert np.array_equal(g["esuel"], fd.esuel) and np.array_equal(g["esuf"], fd.esuf)
        assert np.array_equal(g["inpofa"], fd.inpofa.astype(np.uint64))
        assert np.abs(g["geoFace"] - gf).max() <= 1e-14 and np.abs(g["geoElem"] - ge).max() <= 1e-14
    finally:
        ctx.close()


def test_mesh_from_connectivity_equals_the_hand_assembled_chunk(cases):
    """qdg_mesh_from_connectivity (boundary faces + device FaceData + upload in one call)
    gives the same operator results as the chunk assembled from host FaceData"""
    from quinoa_amd import capi, dgmesh
    case, fix = cases["sedov_dgp1"], load_fixture("sedov_dgp1")
    ss = {int(s): fix["ss_tri_%d" % s] for s in fix["ss_ids"]}
    kw = dict(flux=case["flux"], limiter=case["limiter"], problem=case["problem"], gamma=case["gamma"],
              cfl=case["cfl"], bc_sym=case["bc_sym"], bc_extrapolate=case["bc_extrapolate"])
    ctx = capi.Context(4, **kw)
    try:
        a = dgmesh.upload(ctx, dgmesh.build_chunk(fix["coord"], fix["inpoel"], None, ss))
        b = capi.mesh_from_connectivity(ctx, fix["inpoel"], fix["coord"], ss)
        for m in (a, b):
            m.state_initialize(0.0)
        t = 0.0
        for _ in range(3):
            dta, dtb = a.step(t), b.step(t)
            assert abs(dta - dtb) <= 1e-14 * dta
            t += dta
        Ua, Ub = a.state_download(), b.state_download()
        assert np.abs(Ua - Ub).max() <= 1e-12 * np.abs(Ua).max()
        a.close(); b.close()
    finally:
        ctx.close()


@pytest.mark.parametrize("ndof,limiter,problem", [(4, "superbeep1", "sod_shocktube"), (10, "wenop1", "vortical_flow"),
                                                    (1, "nolimiter", "sod_shocktube")])
def test_device_built_layout_equals_host_built_layout(ndof, limiter, problem):
    """qdg_mesh_from_connectivity builds the whole device layout (Morton order, node / face
    numbering, neighbour and face-code planes, face tasks) on the GPU; option host_layout = 1 routes
    the same call through qdg_mesh_upload's host code.  Same ordering rules => the same mesh:
    stateless operators and a few resident steps agree to rounding (the tile kernel's LDS
    atomics leave last-bit differences), on a mesh with ragged tiles and all six side sets."""
    import os
    from quinoa_amd import capi, meshgen
    ch = meshgen.kuhn_box(11, 9, 7)
    if problem == "sod_shocktube":
        kw = dict(flux="hllc", limiter=limiter, problem=problem, gamma=1.4, cfl=0.3,
                  bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])
    else:
        kw = dict(flux="hllc", limiter=limiter, problem=problem, gamma=5.0 / 3.0, alpha=0.1, beta=1.0, p0=10.0,
                  dt=1e-4, bc_dirichlet=[1, 2, 3, 4, 5, 6])
    res = {}
    for mode in ("device", "host"):
        ctx = capi.Context(ndof, options={"host_layout": 1 if mode == "host" else 0}, **kw)
        mesh = capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"])
        try:
            U0 = mesh.initialize(0.0)
            R = mesh.rhs(0.0, U0)
            L = mesh.lhs()
            mesh.state_upload(U0)
            t = 0.0
            for _ in range(3):
                t += mesh.step(t)
            res[mode] = (U0, R, L, mesh.state_download(), t, mesh.diag(t))
        finally:
            mesh.close(); ctx.close()
    a, b = res["device"], res["host"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    assert np.abs(a[1] - b[1]).max() <= 1e-13 * max(1.0, np.abs(b[1]).max())
    assert abs(a[4] - b[4]) <= 1e-14 * b[4]
    assert np.abs(a[3] - b[3]).max() <= 1e-12 * max(1.0, np.abs(b[3]).max())
    assert np.abs(a[5] - b[5]).max() <= 1e-12 * max(1.0, np.abs(b[5]).max())


def test_device_build_rejects_bad_chunks():
    """qdg_mesh_from_chunk / qdg_mesh_from_connectivity: argument errors and meshes the kernels must
    never see -- an owned tet with a free face that no side set lists -- fail on the host side
    of the call with a message, not in a kernel"""
    from quinoa_amd import capi, meshgen
    ch = meshgen.kuhn_box(2, 2, 2)
    ctx = capi.Context(4, flux="hllc", problem="sod_shocktube", gamma=1.4, cfl=0.3,
                       bc_extrapolate=[1, 2], bc_sym=[3, 4, 5, 6])
    try:
        with pytest.raises(capi.QdgError, match="nielem"):
            capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"], nielem=0)
        with pytest.raises(capi.QdgError, match="nielem"):
            capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"],
                                        nielem=ch["inpoel"].shape[0] + 1)
        # drop one side set: its faces become free faces of owned tets without a boundary entry
        ss = {k: v for k, v in ch["sidesets"].items() if k != 3}
        with pytest.raises(capi.QdgError, match="free face"):
            capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ss)
        # the same tets as GHOSTS may have free faces: only owned tets are checked
        ne = ch["inpoel"].shape[0]
        cen = ch["coord"][ch["inpoel"]].mean(axis=1)
        order = np.argsort(cen[:, 1] > 0.5, kind="stable")          # tets with y < 0.5 first
        nie = int((cen[:, 1] <= 0.5).sum())
        ss2 = {k: v for k, v in ch["sidesets"].items() if k != 4}    # y-max faces belong to the "ghosts"
        m = capi.mesh_from_connectivity(ctx, ch["inpoel"][order], ch["coord"], ss2, nielem=nie)
        assert m.nielem == nie and m.nunk == ne
        m.close()
    finally:
        ctx.close()


def test_overlapping_side_sets():
    """A triangle listed in SEVERAL side sets belongs to the set with the highest id and is integrated once,
    with that set's condition: the reference's loader fills its triangle -> set map set by set in ascending id
    (`faceside[tri] = s.first`, src/Inciter/Partitioner.cpp:358-364 over the std::map m_bface, Partitioner.hpp:198)
    and bndSurfInt finds the face in that one set (src/PDE/Integrate/Boundary.cpp:84-86).  An edge strip of the
    x = 0 side (set 1, Dirichlet) is ALSO listed in set 7 (extrapolate), a strip of the x = 1 side (set 2,
    Dirichlet) also in set 0 (extrapolate; lower id: loses).  The device build = qdg_mesh_upload with the
    reference-style bface = the oracle <= 1e-10; the strips' assignment does matter; children of a doubly
    listed face inherit its one set through the device re-mesh."""
    from oracle import oracle as O
    from quinoa_amd import capi, dgmesh, meshgen
    ch = meshgen.kuhn_box(6, 5, 4)
    coord, inpoel = ch["coord"], ch["inpoel"]
    ss = {int(k): np.asarray(v) for k, v in ch["sidesets"].items()}
    cy = coord[ss[1]].mean(axis=1)[:, 1]
    ss[7] = ss[1][cy < 0.45]                       # the strip of x = 0 next to the edge y = 0
    cy2 = coord[ss[2]].mean(axis=1)[:, 1]
    ss[0] = ss[2][cy2 < 0.45]
    assert 0 < len(ss[7]) < len(ss[1]) and 0 < len(ss[0]) < len(ss[2])
    kw = dict(flux="hllc", limiter="nolimiter", problem="vortical_flow", gamma=5.0 / 3.0, alpha=0.1, beta=1.0, p0=10.0,
              cfl=0.3)
    bcs = dict(bc_dirichlet=[1, 2, 3, 4, 5, 6], bc_extrapolate=[7, 0], bc_sym=[])

    def oracle_for(sidesets):
        om = O.OracleMesh(coord, inpoel, sidesets)
        cfg = O.make_cfg(4, flux="hllc", limiter="nolimiter", problem="vortical_flow", gamma=5.0 / 3.0, alpha=0.1,
                         beta=1.0, p0=10.0)
        return O.Oracle(om, cfg, bcs["bc_dirichlet"], bcs["bc_sym"], bcs["bc_extrapolate"])

    orc = oracle_for(ss)
    Lm = orc.lhs()
    U = orc.initialize(Lm, 0.0)
    t = 0.0
    for _ in range(2):
        t += orc.step(t, U, Lm, cfl=0.3)
    R = orc.rhs(t, U)
    # the assignment matters: with the x = 0 strip left to set 1 (Dirichlet) the residual differs
    plain = {k: v for k, v in ss.items() if k not in (7, 0)}
    Rp = oracle_for(plain).rhs(t, U)
    assert np.abs(Rp - R).max() > 1e-6 * np.abs(R).max()
    # ... and set 0 changes nothing (set 2 has the higher id)
    no0 = {k: v for k, v in ss.items() if k != 0}
    assert np.array_equal(oracle_for(no0).rhs(t, U), R)

    ctx = capi.Context(4, options={"keep_connectivity": 1}, **kw, **bcs)
    dev = capi.mesh_from_connectivity(ctx, inpoel, coord, ss)
    host = dgmesh.upload(ctx, dgmesh.build_chunk(coord, inpoel, None, ss))
    try:
        Rd, Rh = dev.rhs(t, U), host.rhs(t, U)
        scale = max(1.0, np.abs(R).max())
        assert np.abs(Rd - R).max() <= 1e-10 * scale
        assert np.abs(Rh - R).max() <= 1e-10 * scale
        assert np.abs(Rd - Rh).max() <= 1e-12 * scale
        # three resident steps: device-built mesh vs oracle
        dev.state_upload(U)
        Uo, to = U.copy(), t
        for _ in range(3):
            dtg = dev.step(to)
            dto = orc.step(to, Uo, Lm, cfl=0.3)
            assert abs(dtg - dto) <= 1e-12 * dto
            to += dto
        assert np.abs(dev.state_download() - Uo).max() <= 1e-10 * max(1.0, np.abs(Uo).max())
        # the device re-mesh: children of the doubly listed faces carry the winning set
        new, ref = dev.refine_uniform(host_copy=True)
        try:
            c2, i2, ss2, _par = ref.get()
            assert sorted(ss2) == [1, 2, 3, 4, 5, 6, 7]            # set 0 owns no face
            assert len(ss2[7]) == 4 * len(ss[7]) and len(ss2[1]) == 4 * (len(ss[1]) - len(ss[7]))
            om2 = O.OracleMesh(c2, i2, ss2)
            cfg = O.make_cfg(4, flux="hllc", limiter="nolimiter", problem="vortical_flow", gamma=5.0 / 3.0, alpha=0.1,
                             beta=1.0, p0=10.0)
            orc2 = O.Oracle(om2, cfg, bcs["bc_dirichlet"], bcs["bc_sym"], bcs["bc_extrapolate"])
            U2 = orc2.initialize(orc2.lhs(), 0.0)
            R2 = orc2.rhs(0.1, U2)
            assert np.abs(new.rhs(0.1, U2) - R2).max() <= 1e-10 * max(1.0, np.abs(R2).max())
        finally:
            ref.close(); new.close()
    finally:
        dev.close(); host.close(); ctx.close()
