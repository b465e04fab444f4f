"""A general decomposition (qdg_partition + qdg_chunk_build) of the reference's own Sedov
fixture run on the GPU: 4 chunks with ghost halos against the reference's committed 4-PE
baselines (sedov_blastwave_dgp1_pe4.std.exo.{0-3}, sedov_blastwave_pdg_pe4_u0.0.std.exo.{0-3}),
tets matched by centroid, and against the single-chunk GPU run.  All chunks live on the one GPU
of the test box (quinoa_amd.dg.LocalChunks: halo slabs moved by device copies)."""
import numpy as np
import pytest

from conftest import compflow_err, load_fixture
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-10          # north_star bar (the reference's harness: relative 1e-7, exodiff_dg.cfg)


def _centroid_order(c):
    q = np.round(np.asarray(c) * 1e9).astype(np.int64)
    return np.lexsort((q[:, 2], q[:, 1], q[:, 0]))


def _run(case, fix, nparts, method, device_build=False):
    from quinoa_amd import capi, dg, dgmesh, partition
    ss = {int(s): fix["ss_tri_%d" % s] for s in fix["ss_ids"]}
    coord, inpoel = fix["coord"], fix["inpoel"]
    part = partition.partition(coord, inpoel, nparts, method)
    ctx = capi.Context(case["ndof"], flux=case["flux"], limiter=case["limiter"], problem=case["problem"],
                       gamma=case["gamma"], cfl=case["cfl"], dt=case["dt"], bc_dirichlet=case["bc_dirichlet"],
                       bc_sym=case["bc_sym"], bc_extrapolate=case["bc_extrapolate"],
                       pref=case.get("pref", False), tolref=case.get("tolref", 0.1))
    meshes, chunks = [], []
    for r in range(nparts):
        ch = partition.build_chunk(coord, inpoel, ss, part, nparts, r)
        if device_build:
            # FaceData, geometry, ghost-aware device order and face tasks all made on the GPU
            meshes.append(capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"],
                                                      nielem=ch["nielem"]))
        else:
            ck = dgmesh.build_chunk(ch["coord"], ch["inpoel"], ch["nielem"], ch["sidesets"])
            meshes.append(dgmesh.upload(ctx, ck))
        chunks.append(ch)
    try:
        drv = dg.LocalChunks(ctx, meshes, chunks) if nparts > 1 else None
        for m in meshes:
            m.state_initialize(0.0)
        t = 0.0
        for _ in range(case["nstep"]):
            t += drv.step(t) if drv else meshes[0].step(t)
        ne = inpoel.shape[0]
        nf = len(ctx.field_names())
        F = np.zeros((nf, ne))
        for m, ch in zip(meshes, chunks):
            f, names = m.field_output(t)
            F[:, ch["gid"][:ch["nielem"]]] = f
        return F, names, t
    finally:
        for m in meshes:
            m.close()
        ctx.close()


@pytest.mark.parametrize("name,method", [("sedov_dgp1", "rcb"), ("sedov_dgp1", "morton"), ("sedov_pdg", "rcb")])
def test_partitioned_gpu_run_matches_reference_pe4_goldens(name, method, cases):
    case, fix = cases[name], load_fixture(name)
    F4, names, t4 = _run(case, fix, 4, method)
    F1, _, t1 = _run(case, fix, 1, method)
    assert names == [str(n) for n in fix["chunk_names"]]
    scale = np.maximum(1.0, np.abs(F1).max(axis=1))[:, None]
    assert abs(t4 - t1) <= 1e-11 * t1
    assert (np.abs(F4 - F1) / scale).max() <= TOL                       # 4 chunks == 1 chunk
    om = O.OracleMesh(fix["coord"], fix["inpoel"], {})
    cent = om.geoElem.reshape(-1, 4)[:, 1:]
    for tag in (("chunk", "ochunk") if name == "sedov_pdg" else ("chunk",)):
        oa, ob = _centroid_order(cent), _centroid_order(fix[tag + "_centroid"])
        assert np.abs(cent[oa] - fix[tag + "_centroid"][ob]).max() < 1e-12
        gold = fix[tag + "_vals_last"][:, ob]
        assert abs(t4 - float(fix[tag + "_time_last"][0])) <= 1e-10 * t4
        assert (np.abs(F4[:, oa] - gold) / scale).max() <= TOL          # == the reference's 4-PE / 40-chare run
        if name == "sedov_pdg":
            assert np.array_equal(F4[6, oa], gold[6])                   # the per-element ndof field


@pytest.mark.parametrize("name,method", [("sedov_dgp1", "rcb"), ("sedov_pdg", "morton")])
def test_chunks_with_ghosts_built_on_the_device(name, method, cases):
    """qdg_mesh_from_chunk: a rank's chunk WITH its ghost layer from connectivity alone (boundary
    faces of owned tets, chare-boundary faces, halo-adjacent tets last) -- same run as with the
    host-built FaceData + qdg_mesh_upload, and within the bar of the reference's 4-PE baseline"""
    case, fix = cases[name], load_fixture(name)
    Fh, names, th = _run(case, fix, 4, method)
    Fd, names_d, td = _run(case, fix, 4, method, device_build=True)
    assert names == names_d and abs(th - td) <= 1e-13 * th
    scale = np.maximum(1.0, np.abs(Fh).max(axis=1))[:, None]
    assert (np.abs(Fd - Fh) / scale).max() <= 1e-12
    om = O.OracleMesh(fix["coord"], fix["inpoel"], {})
    cent = om.geoElem.reshape(-1, 4)[:, 1:]
    oa, ob = _centroid_order(cent), _centroid_order(fix["chunk_centroid"])
    assert (np.abs(Fd[:, oa] - fix["chunk_vals_last"][:, ob]) / scale).max() <= TOL


@pytest.mark.parametrize("ndof", [4, 10])
def test_device_built_block_chunk_rhs_equals_host_built(ndof):
    """one chunk of the bench's block decomposition (its analytic ghost layer, 3 neighbours):
    stateless RHS, dt and limiter of the device-built mesh against the host-built one on the
    same random state (ghost rows included)"""
    from quinoa_amd import capi, dgmesh, meshgen
    ch = meshgen.kuhn_box_chunk(12, 10, 8, lengths=(1.0, 1.0, 1.0), parts=(2, 2, 2), rank=5)
    lim = "superbeep1" if ndof == 4 else "wenop1"
    ctx = capi.Context(ndof, flux="hllc", limiter=lim, problem="sod_shocktube", gamma=1.4, cfl=0.3,
                       bc_extrapolate=[1, 2], bc_sym=[3, 4, 5, 6])
    mh = dgmesh.upload(ctx, dgmesh.build_chunk(ch["coord"], ch["inpoel"], ch["nielem"], ch["sidesets"]))
    md = capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"], nielem=ch["nielem"])
    try:
        ne = ch["inpoel"].shape[0]
        rng = np.random.default_rng(11)
        U = np.zeros((ne, 5 * ndof))
        U[:, 0] = 1.0 + 0.1 * rng.random(ne)
        U[:, 4 * ndof] = 2.5 + 0.1 * rng.random(ne)
        for c in range(5):
            U[:, c * ndof + 1:(c + 1) * ndof] = 1e-3 * rng.normal(size=(ne, ndof - 1))
        U = U.reshape(-1)
        Rh, Rd = mh.rhs(0.0, U), md.rhs(0.0, U)
        assert np.abs(Rh - Rd).max() <= 1e-13 * max(1.0, np.abs(Rh).max())
        assert abs(mh.dt(U) - md.dt(U)) <= 1e-14 * mh.dt(U)
        assert np.abs(mh.limit(U) - md.limit(U)).max() <= 1e-14
    finally:
        mh.close(); md.close(); ctx.close()


@pytest.mark.parametrize("parts,dims", [((2, 2, 2), (10, 8, 6)), ((2, 2, 1), (9, 8, 5))])
def test_bench_block_decomposition_on_the_gpu_equals_single_chunk(parts, dims):
    """The decomposition bench.py runs on 4 and 8 GPUs (meshgen.kuhn_box_chunk: 2x2x1 and 2x2x2
    blocks, every rank generates its own chunk with its one-layer ghost halo; up to three
    neighbours per rank, uneven segments) through the HIP path: all chunks on this one GPU
    (dg.LocalChunks, device-built chunk meshes, qdg_halo_copy as the transport), Sod DG-P1 +
    Superbee with the CFL time step, 4 steps -- equal to the single chunk run to <= 1e-10."""
    from quinoa_amd import capi, dg, meshgen
    nranks = parts[0] * parts[1] * parts[2]
    kw = dict(flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4, cfl=0.3,
              bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])
    ctx = capi.Context(4, **kw)
    chunks = [meshgen.kuhn_box_chunk(*dims, parts=parts, rank=r) for r in range(nranks)]
    assert max(len(c["nbr_rank"]) for c in chunks) == sum(1 for p in parts if p > 1)
    meshes = [capi.mesh_from_connectivity(ctx, c["inpoel"], c["coord"], c["sidesets"], nielem=c["nielem"])
              for c in chunks]
    one = meshgen.kuhn_box_chunk(*dims, parts=(1, 1, 1), rank=0)
    ctx1 = capi.Context(4, **kw)
    m1 = capi.mesh_from_connectivity(ctx1, one["inpoel"], one["coord"], one["sidesets"])
    try:
        for m in meshes:
            m.state_initialize(0.0)
        m1.state_initialize(0.0)
        drv = dg.LocalChunks(ctx, meshes, chunks)
        t = t1 = 0.0
        for _ in range(4):
            t += drv.step(t)
            t1 += m1.step(t1)
        assert abs(t - t1) <= 1e-12 * t1
        ntet = 6 * dims[0] * dims[1] * dims[2]
        ref = np.zeros((ntet, 20))
        ref[one["gid"][:one["nielem"]]] = m1.state_download().reshape(-1, 20)[:one["nielem"]]
        seen = np.zeros(ntet, dtype=int)
        for c, m in zip(chunks, meshes):
            nie = c["nielem"]
            U = m.state_download().reshape(-1, 20)[:nie]
            g = c["gid"][:nie]
            seen[g] += 1
            assert np.abs(U - ref[g]).max() <= TOL * max(1.0, np.abs(ref).max())
        assert (seen == 1).all()
    finally:
        for m in meshes:
            m.close()
        m1.close(); ctx.close(); ctx1.close()


def _negative_pressure_face_points(U, gamma=1.4):
    """number of face Gauss points (4 faces x 3 points per tet) at which the DG-P1 state U[ne, 20] has a
    non-positive density or pressure -- where HLLC's wave speeds are NaN and its ladder falls through to the
    stored right state (src/PDE/Integrate/Riemann/HLLC.hpp:93-124).  A P1 state is affine: its value at a
    face point is the barycentric mix of its vertex values; Dubiner basis of src/PDE/Integrate/Basis.cpp:267-307."""
    vert = np.array([[0., 0., 0.], [1., 0., 0.], [0., 1., 0.], [0., 0., 1.]])
    B = np.stack([np.ones(4), 2 * vert[:, 0] + vert[:, 1] + vert[:, 2] - 1, 3 * vert[:, 1] + vert[:, 2] - 1,
                  4 * vert[:, 2] - 1], axis=1)                       # [vertex, mode]
    V = np.einsum("eck,vk->ecv", U.reshape(-1, 5, 4), B)              # vertex values
    lpofa = [[1, 2, 3], [2, 0, 3], [3, 0, 1], [0, 2, 1]]
    n = 0
    for f in lpofa:
        for h in range(3):
            w = np.full(3, 1.0 / 6.0); w[h] = 2.0 / 3.0
            s = sum(w[j] * V[:, :, f[j]] for j in range(3))
            with np.errstate(divide="ignore", invalid="ignore"):
                p = (gamma - 1.0) * (s[:, 4] - 0.5 * (s[:, 1] ** 2 + s[:, 2] ** 2 + s[:, 3] ** 2) / s[:, 0])
            n += int(((p <= 0.0) | (s[:, 0] <= 0.0) | ~np.isfinite(p)).sum())
    return n


def test_config4_sedov_block_decomposition_equals_single_chunk_on_every_tet():
    """BASELINE config 4's scheme (Sedov DG-P1 + Superbee, CFL 0.3) on the bench's 2 x 2 x 2 block
    decomposition of a 48^3 box (663 552 tets), 3 CFL steps, all chunks on this GPU: EVERY tet, every
    component, <= 1e-10 of the single-chunk run (tolerance per component, relative to that component's
    largest mean).  The P1 projection of the 10^9 : 1 pressure jump has negative pressure at face points of
    the blast column; HLLC then falls through to the STORED right state (HLLC.hpp:93-124), so the two runs
    agree only because both orient their faces by GLOBAL tet id (qdg_mesh_from_chunk_gid, option
    orient_by_gid).  The test asserts that such face points exist in the limited states the RHS sees at
    the start of the later steps (the limiter keeps the initial projection positive; the blast's first
    steps do not stay so), and that with the chare-local orientation (DG.cpp:480-483; option off) the same decomposition does
    differ -- so it keeps testing the fall-through."""
    from quinoa_amd import capi, dg, meshgen
    parts, nx, nsteps = (2, 2, 2), 48, 3
    kw = dict(flux="hllc", limiter="superbeep1", problem="sedov_blastwave", gamma=1.4, cfl=0.3,
              bc_extrapolate=[2, 4], bc_sym=[1, 3, 5, 6])
    ntet = 6 * nx ** 3
    one = meshgen.kuhn_box_chunk(nx, nx, nx, parts=(1, 1, 1), rank=0)
    chunks = [meshgen.kuhn_box_chunk(nx, nx, nx, parts=parts, rank=r) for r in range(8)]

    def single():
        ctx = capi.Context(4, **kw)
        m = capi.mesh_from_connectivity(ctx, one["inpoel"], one["coord"], one["sidesets"], elem_gid=one["gid"])
        try:
            m.state_initialize(0.0)
            t, nneg = 0.0, []
            for _ in range(nsteps):
                t += m.step(t)
                # the limited state the next stage-0 RHS sees
                nneg.append(_negative_pressure_face_points(m.limit(m.state_download()).reshape(-1, 20)))
            ref = np.zeros((ntet, 20))
            ref[one["gid"]] = m.state_download().reshape(-1, 20)
            return ref, t, nneg
        finally:
            m.close(); ctx.close()

    def decomposed(orient_by_gid):
        ctx = capi.Context(4, **kw)
        ctx.set_option("orient_by_gid", int(orient_by_gid))
        meshes = [capi.mesh_from_connectivity(ctx, c["inpoel"], c["coord"], c["sidesets"], nielem=c["nielem"],
                                              elem_gid=c["gid"]) for c in chunks]
        try:
            for m in meshes:
                m.state_initialize(0.0)
            drv = dg.LocalChunks(ctx, meshes, chunks)
            t = 0.0
            for _ in range(nsteps):
                t += drv.step(t)
            out = np.zeros((ntet, 20))
            seen = np.zeros(ntet, dtype=int)
            for c, m in zip(chunks, meshes):
                nie = c["nielem"]
                out[c["gid"][:nie]] = m.state_download().reshape(-1, 20)[:nie]
                seen[c["gid"][:nie]] += 1
            assert (seen == 1).all()
            return out, t
        finally:
            for m in meshes:
                m.close()
            ctx.close()

    ref, t1, nneg = single()
    assert sum(nneg) > 0, "no face point with p <= 0 (%r): the test no longer exercises HLLC's fall-through" % (nneg,)
    got, t = decomposed(True)
    assert abs(t - t1) <= 1e-13 * t1
    err = compflow_err(got, ref, 4)               # per component, max over ALL tets and DOFs
    assert err <= TOL, "worst tet %d: %.2e" % (int(np.abs(got - ref).max(axis=1).argmax()), err)
    loc, _ = decomposed(False)
    assert compflow_err(loc, ref, 4) > 10 * TOL, \
        "chare-local orientation no longer differs: is the fall-through still reached?"


@pytest.mark.parametrize("device_build", [True, False])
def test_general_partition_with_global_ids_equals_the_serial_run(device_build):
    """The serial run of a mesh (one chunk, NO global ids: left tet of a face = lower tet id, the
    reference's own rule, DerivedData.cpp:1127-1139) against the same mesh cut into 5 RCB chunks by
    qdg_partition / qdg_chunk_build whose meshes are built with the chunks' elem_gid -- through the device
    build (qdg_mesh_from_chunk_gid) and through the host-array upload (qdg_mesh_upload_gid, the caller's
    FaceData keeps left = owned tet).  Sedov on a 20^3 box, 3 CFL steps, every tet <= 1e-10."""
    from quinoa_amd import capi, dg, dgmesh, meshgen, partition
    kw = dict(flux="hllc", limiter="superbeep1", problem="sedov_blastwave", gamma=1.4, cfl=0.3,
              bc_extrapolate=[2, 4], bc_sym=[1, 3, 5, 6])
    g = meshgen.kuhn_box(20, 20, 20)
    coord, inpoel, ss = g["coord"], g["inpoel"], g["sidesets"]
    ctx1 = capi.Context(4, **kw)
    m1 = capi.mesh_from_connectivity(ctx1, inpoel, coord, ss)
    ctx = capi.Context(4, **kw)
    nparts = 5
    part = partition.partition(coord, inpoel, nparts, "rcb")
    chunks = [partition.build_chunk(coord, inpoel, ss, part, nparts, r) for r in range(nparts)]
    if device_build:
        meshes = [capi.mesh_from_connectivity(ctx, c["inpoel"], c["coord"], c["sidesets"], nielem=c["nielem"],
                                              elem_gid=c["gid"]) for c in chunks]
    else:
        meshes = [dgmesh.upload(ctx, dgmesh.build_chunk(c["coord"], c["inpoel"], c["nielem"], c["sidesets"]),
                                elem_gid=c["gid"]) for c in chunks]
    try:
        m1.state_initialize(0.0)
        for m in meshes:
            m.state_initialize(0.0)
        drv = dg.LocalChunks(ctx, meshes, chunks)
        t = t1 = 0.0
        for _ in range(3):
            t += drv.step(t)
            t1 += m1.step(t1)
        assert abs(t - t1) <= 1e-13 * t1
        ref = m1.state_download().reshape(-1, 20)
        got = np.zeros_like(ref)
        for c, m in zip(chunks, meshes):
            nie = c["nielem"]
            got[c["gid"][:nie]] = m.state_download().reshape(-1, 20)[:nie]
        assert compflow_err(got, ref, 4) <= TOL
    finally:
        for m in meshes:
            m.close()
        m1.close(); ctx.close(); ctx1.close()


@pytest.mark.parametrize("ndof,limiter", [(4, "superbeep1"), (10, "wenop1")])
def test_multi_scalar_transport_on_a_decomposition_equals_single_chunk(ndof, limiter):
    """dg::Transport with three scalars (rows of 3 * ndof doubles) in 2 x 2 x 1 chunks with ghost halos on
    the one GPU: the halo rows carry all scalars; equal to the single-chunk run."""
    from quinoa_amd import capi, dg, meshgen
    parts, dims = (2, 2, 1), (8, 7, 5)
    kw = dict(pde="transport", flux="upwind", problem="slot_cyl", dt=2.0e-3, limiter=limiter, ncomp=3,
              bc_dirichlet=[1, 2, 3, 4], bc_extrapolate=[5, 6])
    ctx = capi.Context(ndof, **kw)
    chunks = [meshgen.kuhn_box_chunk(*dims, parts=parts, rank=r) for r in range(4)]
    meshes = [capi.mesh_from_connectivity(ctx, c["inpoel"], c["coord"], c["sidesets"], nielem=c["nielem"])
              for c in chunks]
    one = meshgen.kuhn_box_chunk(*dims, parts=(1, 1, 1), rank=0)
    ctx1 = capi.Context(ndof, **kw)
    m1 = capi.mesh_from_connectivity(ctx1, one["inpoel"], one["coord"], one["sidesets"])
    np_ = 3 * ndof
    try:
        assert m1.nprop == np_
        for m in meshes:
            m.state_initialize(0.0)
        m1.state_initialize(0.0)
        drv = dg.LocalChunks(ctx, meshes, chunks)
        t = t1 = 0.0
        for _ in range(3):
            t += drv.step(t)
            t1 += m1.step(t1)
        assert abs(t - t1) <= 1e-14
        ntet = 6 * dims[0] * dims[1] * dims[2]
        ref = np.zeros((ntet, np_))
        ref[one["gid"][:one["nielem"]]] = m1.state_download().reshape(-1, np_)[:one["nielem"]]
        for c, m in zip(chunks, meshes):
            nie = c["nielem"]
            U = m.state_download().reshape(-1, np_)[:nie]
            assert np.abs(U - ref[c["gid"][:nie]]).max() <= TOL * max(1.0, np.abs(ref).max())
        assert np.abs(ref[:, ndof]).max() > 0.0          # scalar 1 is there
    finally:
        for m in meshes:
            m.close()
        m1.close(); ctx.close(); ctx1.close()
