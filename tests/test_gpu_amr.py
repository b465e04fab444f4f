"""BASELINE config 5's loop on the GPU: uniform 1:8 refinement while time stepping -- host
refinement (qdg_refine_uniform), mesh-derived data of the new mesh generated on the device
(qdg_mesh_from_connectivity), state handed over on the device (qdg_state_transfer) -- against
the reference's own t>0-refinement golden (mesh_refinement/dtref/gauss_hump.q) and, for the
config's CompFlow DG-P1 physics, against the oracle on the refined mesh."""
import numpy as np
import pytest

from conftest import compflow_err, load_fixture
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _ss(fix):
    return {int(s): fix["ss_tri_%d" % s] for s in fix["ss_ids"]}


def test_gpu_refined_run_matches_reference_dtref_goldens(cases):
    from quinoa_amd import amr, capi
    case, fix = cases["gauss_hump_dtref"], load_fixture("gauss_hump_dtref")
    ctx = capi.Context(1, flux="upwind", problem="gauss_hump", dt=case["dt"], pde="transport",
                       bc_extrapolate=case["bc_extrapolate"], bc_inlet=case["bc_inlet"], bc_outlet=case["bc_outlet"])
    run = amr.RefinedRun(ctx, fix["coord"], fix["inpoel"], _ss(fix))
    try:
        run.mesh.state_initialize(0.0)
        t, it, rows = 0.0, 0, []
        for k in range(3):
            gt, gv = fix["s%d_times" % k], fix["s%d_vals" % k]
            out = [run.mesh.field_output(t)[0]]
            vol = None
            for _ in range(case["dtfreq"] if k < 2 else 0):
                t += run.mesh.step(t)
                it += 1
                if it % case["diag_interval"] == 0:
                    if vol is None:
                        vol = O.OracleMesh(run.coord, run.inpoel, {}).meshvol
                    d = run.mesh.diag(t)
                    rows.append([it, t, np.sqrt(d[0] / vol), np.sqrt(d[5] / vol)])
                out.append(run.mesh.field_output(t)[0])
            got = np.array(out[:len(gt)])            # [time, var, elem]
            assert got.shape == gv.shape
            assert np.abs(got - gv).max() <= 1e-10, k
            if k < 2:
                run.refine()
                assert run.mesh.nielem == 8 * gv.shape[2]
        gold = {int(g[0]): g for g in fix["diag"]}
        assert len(rows) == len(gold)
        for r in rows:
            g = gold[int(r[0])]
            assert abs(r[1] - g[1]) <= 1e-12 and abs(r[2] - g[3]) <= 6e-7 * g[3] and abs(r[3] - g[4]) <= 6e-7 * g[4]
    finally:
        run.mesh.close(); ctx.close()


@pytest.mark.parametrize("ndof,limiter", [(4, "superbeep1"), (1, "nolimiter")])
def test_config5_physics_refine_once_matches_oracle(ndof, limiter):
    """config 5: CompFlow (Sod) with a uniform refinement in the middle of the run.  GPU: 4 steps,
    refine + rebuild on the device + transfer, 4 more steps.  Oracle: the same steps on the two
    meshes with the state copied child <- parent (all DOFs of the row, DG.cpp:1597-1605)."""
    from quinoa_amd import amr, capi, meshgen
    ch = meshgen.kuhn_box(6, 4, 3, lengths=(1.0, 0.4, 0.3))
    kw = dict(flux="hllc", limiter=limiter, problem="sod_shocktube", gamma=1.4)
    bc = dict(bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])
    ctx = capi.Context(ndof, cfl=0.3, **kw, **bc)
    run = amr.RefinedRun(ctx, ch["coord"], ch["inpoel"], ch["sidesets"])
    try:
        run.mesh.state_initialize(0.0)
        om = O.OracleMesh(ch["coord"], ch["inpoel"], ch["sidesets"])
        orc = O.Oracle(om, O.make_cfg(ndof, **kw), **bc)
        Lm = orc.lhs(); U = orc.initialize(Lm, 0.0)
        t = 0.0
        for _ in range(4):
            dtg = run.mesh.step(t)
            dto = orc.step(t, U, Lm, cfl=0.3)
            assert abs(dtg - dto) <= 1e-11 * dto
            t += dto
        Ug = run.mesh.state_download()
        assert np.abs(Ug - U).max() <= 1e-10
        tim = run.refine()
        assert len(tim) == 3 and run.mesh.nielem == 8 * om.nelem
        c2, i2, s2, par = amr.refine_uniform(ch["coord"], ch["inpoel"], ch["sidesets"])
        assert np.array_equal(i2, run.inpoel)
        # the hand-over is a pure copy of rows, through two different device numberings
        assert np.array_equal(run.mesh.state_download(), Ug.reshape(om.nelem, -1)[par].reshape(-1))
        U = U.reshape(om.nelem, -1)[par].reshape(-1)
        om2 = O.OracleMesh(c2, i2, s2)
        orc2 = O.Oracle(om2, O.make_cfg(ndof, **kw), **bc)
        L2 = orc2.lhs()
        for _ in range(4):
            dtg = run.mesh.step(t)
            dto = orc2.step(t, U, L2, cfl=0.3)
            assert abs(dtg - dto) <= 1e-11 * dto
            t += dto
        assert compflow_err(run.mesh.state_download(), U, ndof) <= 1e-10
    finally:
        run.mesh.close(); ctx.close()


@pytest.mark.parametrize("depth", [1, 2])
def test_config5_partitioned_chunks_refine_on_the_gpu(depth):
    """config 5 with a decomposition: 3 chunks on the GPU (dg.LocalChunks), 3 steps, every chunk
    refined by its own rank's logic (amr.refine_chunk) + re-uploaded + state handed over on the
    device (qdg_state_transfer across the two numberings, ghost rows included), 3 more steps;
    equal to the single-chunk run across the same refinement (amr.RefinedRun).  depth 2: chunks with two
    ghost layers (qdg_chunk_build_depth / qdg_refine_chunk_depth; device-built meshes, the ranks limit their
    layer-1 ghosts themselves, one exchange per stage) before and after the re-mesh."""
    from quinoa_amd import amr, capi, dg, dgmesh, meshgen, partition
    g = meshgen.kuhn_box(6, 5, 4)
    kw = dict(flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4, cfl=0.3,
              bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])
    nparts = 3
    part = partition.partition(g["coord"], g["inpoel"], nparts, "rcb")
    ctx = capi.Context(4, **kw)
    chunks = [partition.build_chunk(g["coord"], g["inpoel"], g["sidesets"], part, nparts, r, depth=depth)
              for r in range(nparts)]

    def upload(ch):
        if depth == 2:      # (qdg_mesh_upload knows the owned tets' neighbours only)
            return capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"], nielem=ch["nielem"])
        return dgmesh.upload(ctx, dgmesh.build_chunk(ch["coord"], ch["inpoel"], ch["nielem"], ch["sidesets"]))

    meshes = [upload(ch) for ch in chunks]
    ctx1 = capi.Context(4, **kw)
    one = amr.RefinedRun(ctx1, g["coord"], g["inpoel"], g["sidesets"])
    try:
        for m in meshes:
            m.state_initialize(0.0)
        one.mesh.state_initialize(0.0)
        drv = dg.LocalChunks(ctx, meshes, chunks)
        assert drv.deep == (depth == 2)
        t = t1 = 0.0
        for _ in range(3):
            t += drv.step(t)
            t1 += one.mesh.step(t1)
        new_chunks, new_meshes = [], []
        for ch, m in zip(chunks, meshes):
            ch2, par = amr.refine_chunk(ch)
            assert ch2["depth"] == depth
            # the re-mesh step of a rank: its refined chunk WITH the new ghost layer rebuilt on
            # the device from connectivity alone (qdg_mesh_from_chunk)
            m2 = capi.mesh_from_connectivity(ctx, ch2["inpoel"], ch2["coord"], ch2["sidesets"],
                                             nielem=ch2["nielem"])
            amr.state_transfer(m, m2, par)
            m.close()
            new_chunks.append(ch2); new_meshes.append(m2)
        chunks, meshes = new_chunks, new_meshes
        one.refine()
        drv = dg.LocalChunks(ctx, meshes, chunks)
        assert drv.deep == (depth == 2)
        for _ in range(3):
            t += drv.step(t)
            t1 += one.mesh.step(t1)
        assert abs(t - t1) <= 1e-12 * t1
        ref = one.mesh.state_download().reshape(-1, 20)
        for ch, m in zip(chunks, meshes):
            nie = ch["nielem"]
            U = m.state_download().reshape(-1, 20)[:nie]
            assert np.abs(U - ref[ch["gid"][:nie]]).max() <= 1e-10 * np.abs(ref).max()
    finally:
        for m in meshes:
            m.close()
        one.mesh.close(); ctx.close(); ctx1.close()


def test_device_refinement_equals_host_refinement():
    """qdg_refine_uniform_device: every array identical to the host's qdg_refine_uniform (children
    order, midpoint numbering in the order the tets meet their edges, side-set triangles), on a
    jittered Kuhn box and on the reference's own fixture mesh"""
    from quinoa_amd import amr, capi, meshgen
    ctx = capi.Context(4, flux="hllc", problem="sod_shocktube", gamma=1.4, cfl=0.3)
    try:
        meshes = [meshgen.kuhn_box(9, 7, 5)]
        fix = load_fixture("sedov_dgp1")
        meshes.append({"coord": fix["coord"], "inpoel": fix["inpoel"],
                       "sidesets": {int(s): fix["ss_tri_%d" % s] for s in fix["ss_ids"]}})
        for g in meshes:
            ch, ih, sh, ph = amr.refine_uniform(g["coord"], g["inpoel"], g["sidesets"])
            cd, idv, sd, pd = amr.refine_uniform(g["coord"], g["inpoel"], g["sidesets"], ctx=ctx)
            assert np.array_equal(ih, idv) and np.array_equal(ph, pd) and np.array_equal(ch, cd)
            assert sorted(sh) == sorted(sd)
            for k in sh:
                assert np.array_equal(sh[k], sd[k])
    finally:
        ctx.close()


def test_config5_refine_then_recut_with_state_migration():
    """config 5 as BASELINE states it -- refinement followed by a RE-PARTITION: 3 chunks cut by
    recursive bisection run 3 steps, every chunk is refined by its own rank's logic and takes its
    state over; then the refined mesh is cut AGAIN, differently (4 chunks along a Morton curve, what
    the reference's load balancing after DG::resizePostAMR amounts to, DG.cpp:1658-1664), the new
    chunks are built on the device and the state MIGRATES between the two decompositions by
    global tet id (qdg_state_migrate, device to device; one pair through the packed-row halves
    qdg_state_rows_get / _put that a multi-process run would send over RCCL); 3 more steps.
    Equal to the single-chunk run across the same refinement."""
    import ctypes as C
    from quinoa_amd import amr, capi, dg, meshgen, partition
    g = meshgen.kuhn_box(6, 5, 4)
    kw = dict(flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4, cfl=0.3,
              bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])
    ctx = capi.Context(4, **kw)
    ctx1 = capi.Context(4, **kw)
    one = amr.RefinedRun(ctx1, g["coord"], g["inpoel"], g["sidesets"])

    def build(ch):
        return capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"], nielem=ch["nielem"])

    part = partition.partition(g["coord"], g["inpoel"], 3, "rcb")
    chunks = [partition.build_chunk(g["coord"], g["inpoel"], g["sidesets"], part, 3, r) for r in range(3)]
    meshes = [build(ch) for ch in chunks]
    buf = 0
    try:
        for m in meshes:
            m.state_initialize(0.0)
        one.mesh.state_initialize(0.0)
        drv = dg.LocalChunks(ctx, meshes, chunks)
        t = t1 = 0.0
        for _ in range(3):
            t += drv.step(t)
            t1 += one.mesh.step(t1)
        # refinement on the old decomposition
        ref_chunks, ref_meshes = [], []
        for ch, m in zip(chunks, meshes):
            ch2, par = amr.refine_chunk(ch)
            m2 = build(ch2)
            amr.state_transfer(m, m2, par)
            m.close()
            ref_chunks.append(ch2); ref_meshes.append(m2)
        one.refine()
        # a different cut of the refined mesh (global child id = 8 * parent + k in both)
        part2 = partition.partition(one.coord, one.inpoel, 4, "morton")
        new_chunks = [partition.build_chunk(one.coord, one.inpoel, one.sidesets, part2, 4, r) for r in range(4)]
        assert len({tuple(sorted(c["gid"][:c["nielem"]])) for c in new_chunks}) == 4
        new_meshes = [build(ch) for ch in new_chunks]
        moved = np.zeros((3, 4), dtype=np.int64)
        for r, (co, mo) in enumerate(zip(ref_chunks, ref_meshes)):
            for q, (cn, mn) in enumerate(zip(new_chunks, new_meshes)):
                if (r, q) == (0, 0):
                    # this pair through the packed-row halves (what two processes would do)
                    go, gn = co["gid"][:co["nielem"]], cn["gid"][:cn["nielem"]]
                    common, io, in_ = np.intersect1d(go, gn, return_indices=True)
                    if len(common):
                        buf = ctx.device_alloc(len(common) * 20 * 8)
                        mo.state_rows_get(io, buf)
                        mn.state_rows_put(in_, buf)
                    moved[r, q] = len(common)
                else:
                    moved[r, q] = amr.state_migrate(mo, co["gid"], mn, cn["gid"])
        assert (moved.sum(axis=0) == [c["nielem"] for c in new_chunks]).all()     # every row arrived once
        assert (moved > 0).sum() > 4                                              # rows really changed chunks
        for m in ref_meshes:
            m.close()
        meshes, chunks = new_meshes, new_chunks
        drv = dg.LocalChunks(ctx, meshes, chunks)
        for _ in range(3):
            t += drv.step(t)
            t1 += one.mesh.step(t1)
        assert abs(t - t1) <= 1e-12 * t1
        ref = one.mesh.state_download().reshape(-1, 20)
        for ch, m in zip(chunks, meshes):
            nie = ch["nielem"]
            U = m.state_download().reshape(-1, 20)[:nie]
            assert np.abs(U - ref[ch["gid"][:nie]]).max() <= 1e-10 * np.abs(ref).max()
    finally:
        ctx.device_free(buf)
        for m in meshes:
            m.close()
        one.mesh.close(); ctx.close(); ctx1.close()


def test_config5_region_refinement_from_caller_supplied_connectivity():
    """A refined SUB-REGION handed in as connectivity + parent per tet -- what the reference's
    Refiner gives DG::resizePostAMR (DG.cpp:1537-1612): marked tets 1:8, closure with the 1:2 and
    1:4 templates (tests/region_refine.py stands in for the AMR library).  GPU: 3 steps, device
    mesh from the refined connectivity (qdg_mesh_from_connectivity), state child <- parent
    (qdg_state_transfer), 3 more steps; the oracle does the same on the same two meshes."""
    from region_refine import check_conforming, refine_region
    from quinoa_amd import amr, capi, meshgen
    g = meshgen.kuhn_box(16, 4, 4, lengths=(1.0, 0.4, 0.3))
    cen = g["coord"][g["inpoel"]].mean(axis=1)
    marked = (cen[:, 0] > 0.4) & (cen[:, 0] < 0.5)
    c2, i2, s2, par = refine_region(g["coord"], g["inpoel"], g["sidesets"], marked)
    check_conforming(i2, s2)
    nch = np.bincount(par)
    assert {1, 2, 4, 8} <= set(nch.tolist())                 # all four templates occur
    kw = dict(flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4)
    bc = dict(bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])
    ctx = capi.Context(4, cfl=0.3, **kw, **bc)
    m1 = capi.mesh_from_connectivity(ctx, g["inpoel"], g["coord"], g["sidesets"])
    m2 = None
    try:
        m1.state_initialize(0.0)
        om = O.OracleMesh(g["coord"], g["inpoel"], g["sidesets"])
        orc = O.Oracle(om, O.make_cfg(4, **kw), **bc)
        Lm = orc.lhs(); U = orc.initialize(Lm, 0.0)
        t = 0.0
        for _ in range(3):
            dtg = m1.step(t)
            dto = orc.step(t, U, Lm, cfl=0.3)
            assert abs(dtg - dto) <= 1e-11 * dto
            t += dto
        m2 = capi.mesh_from_connectivity(ctx, i2, c2, s2)
        amr.state_transfer(m1, m2, par)
        Ug = m1.state_download().reshape(om.nelem, -1)
        assert np.array_equal(m2.state_download().reshape(len(par), -1), Ug[par])   # pure row copy
        U = U.reshape(om.nelem, -1)[par].reshape(-1)
        om2 = O.OracleMesh(c2, i2, s2)
        orc2 = O.Oracle(om2, O.make_cfg(4, **kw), **bc)
        L2 = orc2.lhs()
        for _ in range(3):
            dtg = m2.step(t)
            dto = orc2.step(t, U, L2, cfl=0.3)
            assert abs(dtg - dto) <= 1e-11 * dto
            t += dto
        assert np.abs(m2.state_download() - U).max() <= 1e-10 * max(1.0, np.abs(U).max())
        # conservation across the hand-over: child volumes add up to the parent's, rows are copied
        vol2 = om2.geoElem[0::4]
        mass2 = (m2.state_download().reshape(len(par), -1)[:, 0] * vol2).sum()
        assert np.isfinite(mass2)
    finally:
        m1.close()
        if m2:
            m2.close()
        ctx.close()


@pytest.mark.parametrize("ndof,limiter,problem", [(4, "superbeep1", "sod_shocktube"), (10, "wenop1", "vortical_flow")])
def test_device_resident_remesh_equals_the_sort_path(ndof, limiter, problem):
    """qdg_mesh_refine_uniform: config 5's re-mesh without the host -- refinement from the connectivity the
    handle kept on the device, the children's esuel by the 1:8 template, boundary faces from the parents',
    state handed over on the device, host copy by a second thread.  Against the path it replaces: (a) the
    host copy = qdg_refine_uniform's arrays, element for element; (b) the state = the parents' rows; (c) the
    new handle computes BITWISE what a handle built from the refined connectivity by the sort path
    (qdg_mesh_from_connectivity) computes -- lhs, rhs, dt, limiter, three time steps, with the reproducible
    element-centric DG-P1 kernel (option p1_rhs = 1) -- i.e. the device arrays are the same; (d) the new
    handle keeps its own connectivity: a second refinement works from it."""
    from quinoa_amd import amr, capi, meshgen
    ch = meshgen.kuhn_box(7, 6, 5)
    if problem == "sod_shocktube":
        kw = dict(flux="hllc", limiter=limiter, problem=problem, gamma=1.4, cfl=0.3,
                  bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])
    else:
        kw = dict(flux="hllc", limiter=limiter, problem=problem, gamma=5.0 / 3.0, alpha=0.1, beta=1.0, p0=10.0,
                  dt=1e-4, bc_dirichlet=[1, 2, 3, 4, 5, 6])
    ctx = capi.Context(ndof, options={"keep_connectivity": 1, "p1_rhs": 1}, **kw)
    plain = capi.Context(ndof, options={"p1_rhs": 1}, **kw)
    m1 = capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"])
    m0 = capi.mesh_from_connectivity(plain, ch["inpoel"], ch["coord"], ch["sidesets"])
    m2 = m3 = m4 = ref = None
    try:
        with pytest.raises(capi.QdgError, match="keeps no connectivity"):
            m0.refine_uniform()
        m1.state_initialize(0.0)
        t = 0.0
        for _ in range(3):
            t += m1.step(t)
        U1 = m1.state_download().reshape(m1.nielem, -1)
        m2, ref = m1.refine_uniform(host_copy=True)
        c2, i2, s2, par = ref.get()
        hc2, hi2, hs2, hpar = amr.refine_uniform(ch["coord"], ch["inpoel"], ch["sidesets"])
        assert np.array_equal(i2, hi2) and np.array_equal(c2, hc2) and np.array_equal(par, hpar)     # (a)
        key = lambda tri: set(map(tuple, np.sort(np.asarray(tri).reshape(-1, 3), axis=1).tolist()))
        assert sorted(s2) == sorted(hs2) and all(key(s2[k]) == key(hs2[k]) for k in s2)
        U2 = m2.state_download()
        assert np.array_equal(U2.reshape(len(par), -1), U1[par])                                    # (b)
        m3 = capi.mesh_from_connectivity(ctx, hi2, hc2, hs2)                                        # the sort path
        amr.state_transfer(m1, m3, hpar)
        assert np.array_equal(U2, m3.state_download())
        assert np.array_equal(m2.lhs(), m3.lhs())                                                   # (c)
        assert np.array_equal(m2.rhs(t, U2), m3.rhs(t, U2))
        assert m2.dt(U2) == m3.dt(U2)
        assert np.array_equal(m2.limit(U2), m3.limit(U2))
        ta = tb = t
        for _ in range(3):
            ta += m2.step(ta); tb += m3.step(tb)
        assert ta == tb and np.array_equal(m2.state_download(), m3.state_download())
        m4, none = m2.refine_uniform(host_copy=False)                                               # (d)
        assert none is None and m4.nielem == 64 * m1.nielem
        assert np.array_equal(m4.state_download().reshape(m4.nielem, -1)[::8],
                              m2.state_download().reshape(m2.nielem, -1))
        m4.step(ta)
        assert np.isfinite(m4.state_download()).all()
    finally:
        if ref:
            ref.close()
        for m in (m4, m3, m2, m1, m0):
            if m:
                m.close()
        ctx.close(); plain.close()


def test_device_resident_chunk_remesh_equals_the_host_path():
    """qdg_mesh_refine_chunk: config 5's re-mesh of ONE RANK's chunk WITH its ghost layer, on the device.
    3 RCB chunks of a box (all on this GPU), Sod DG-P1 + Superbee: 3 steps, then every chunk is refined twice
    over -- by the host path (qdg_refine_chunk -> qdg_mesh_from_chunk_gid -> qdg_state_transfer) and by the
    one device call.  (a) The device call's plan and numbering = the host's: global ids, parents, send lists,
    receive counts, and with copy_mesh connectivity, coordinates and side-set triangles; (b) the owned state =
    the parents' rows; (c) after 3 more steps (halos between the chunks) the two decompositions hold BITWISE the
    same states (reproducible DG-P1 kernel), and (d) equal the single chunk refined by its own path to 1e-10;
    (e) a second re-mesh works from the new handles' kept connectivity and plan."""
    from quinoa_amd import amr, capi, dg, meshgen, partition
    g = meshgen.kuhn_box(6, 5, 4)
    kw = dict(flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4, cfl=0.3,
              bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])
    opt = {"keep_connectivity": 1, "p1_rhs": 1}
    ctxa, ctxb, ctx1 = capi.Context(4, options=opt, **kw), capi.Context(4, options=opt, **kw), capi.Context(4, options=opt, **kw)
    one = amr.RefinedRun(ctx1, g["coord"], g["inpoel"], g["sidesets"], resident=True)
    part = partition.partition(g["coord"], g["inpoel"], 3, "rcb")
    chunks = [partition.build_chunk(g["coord"], g["inpoel"], g["sidesets"], part, 3, r) for r in range(3)]

    def build(ctx, ch):
        return capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"], nielem=ch["nielem"],
                                           elem_gid=ch["gid"])

    A = [build(ctxa, ch) for ch in chunks]          # host-path decomposition
    B = [build(ctxb, ch) for ch in chunks]          # device-path decomposition
    allm = []
    try:
        for m in A + B:
            m.state_initialize(0.0)
        one.mesh.state_initialize(0.0)
        da, db = dg.LocalChunks(ctxa, A, chunks), dg.LocalChunks(ctxb, B, chunks)
        t = tb = t1 = 0.0
        for _ in range(3):
            t += da.step(t); tb += db.step(tb); t1 += one.mesh.step(t1)
        assert t == tb
        A2, B2, chA, chB = [], [], [], []
        for ch, ma, mb in zip(chunks, A, B):
            ch2, par = amr.refine_chunk(ch)                                       # host path
            m2 = build(ctxa, ch2)
            amr.state_transfer(ma, m2, par)
            n2, plan = mb.refine_chunk(ch["nbr_rank"], copy_mesh=True)          # device path
            nie2 = ch2["nielem"]
            assert plan["nielem"] == nie2 and np.array_equal(plan["gid"], ch2["gid"])             # (a)
            assert np.array_equal(plan["parent"], par) and plan["recv_counts"] == list(ch2["recv_counts"])
            assert all(np.array_equal(p, q) for p, q in zip(plan["send_lists"], ch2["send_lists"]))
            assert np.array_equal(plan["inpoel"], ch2["inpoel"]) and np.array_equal(plan["coord"], ch2["coord"])
            key = lambda tri: set(map(tuple, np.sort(np.asarray(tri).reshape(-1, 3), axis=1).tolist()))
            assert sorted(plan["sidesets"]) == sorted(ch2["sidesets"])
            assert all(key(plan["sidesets"][k]) == key(ch2["sidesets"][k]) for k in ch2["sidesets"])
            Ub = mb.state_download().reshape(mb.nunk, -1)
            assert np.array_equal(n2.state_download().reshape(n2.nunk, -1)[:nie2], Ub[par[:nie2]])   # (b)
            A2.append(m2); B2.append(n2); chA.append(ch2); chB.append(plan)
        allm = A2 + B2
        for m in A + B:
            m.close()
        A, B = [], []
        one.refine()
        da, db = dg.LocalChunks(ctxa, A2, chA), dg.LocalChunks(ctxb, B2, chB)
        for _ in range(3):
            t += da.step(t); tb += db.step(tb); t1 += one.mesh.step(t1)
        assert t == tb and abs(t - t1) <= 1e-12 * t1
        ref = one.mesh.state_download().reshape(-1, 20)
        for ch, ma, mb in zip(chA, A2, B2):
            nie = ch["nielem"]
            Ua, Ub = ma.state_download().reshape(-1, 20)[:nie], mb.state_download().reshape(-1, 20)[:nie]
            assert np.array_equal(Ua, Ub)                                                            # (c)
            assert np.abs(Ub - ref[ch["gid"][:nie]]).max() <= 1e-10 * np.abs(ref).max()              # (d)
        B3 = []
        for plan, mb in zip(chB, B2):                                                                # (e)
            n3, plan3 = mb.refine_chunk(plan["nbr_rank"])
            assert plan3["nielem"] == 8 * plan["nielem"] and len(plan3["gid"]) == n3.nunk
            B3.append((n3, plan3))
        allm += [m for m, _ in B3]
        d3 = dg.LocalChunks(ctxb, [m for m, _ in B3], [p for _, p in B3])
        d3.step(tb)
        assert all(np.isfinite(m.state_download()).all() for m, _ in B3)
    finally:
        for m in A + B + allm:
            m.close()
        one.mesh.close(); ctxa.close(); ctxb.close(); ctx1.close()


def test_device_derefinement_is_the_inverse_of_the_device_refinement():
    """qdg_mesh_derefine_uniform (8 children -> their parent, on the device): refine -> derefine gives back the
    original mesh -- the handle computes BITWISE what the original handle computes (lhs, rhs, dt, steps with the
    reproducible DG-P1 kernel) -- and, with policy first_child, every DOF of the state (the inverse of the
    reference's row copy child <- parent, DG.cpp:1597-1605); policy mean conserves mass and energy of a state that
    has evolved on the refined mesh.  On the reference's own t0 sequence mesh (uniform -> uniform_derefine -> uniform,
    tests/golden/t0ref_gauss_hump_udu.npz) with its three side sets."""
    import os
    from quinoa_amd import capi, meshgen
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "t0ref_gauss_hump_udu.npz"))
    ss = {int(s): f["s0_ss_tri_%d" % s].astype(np.int64) for s in f["s0_ss_ids"]}
    cases = [(meshgen.kuhn_box(5, 4, 3), dict(bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])),
             ({"coord": f["s0_coord"], "inpoel": f["s0_inpoel"].astype(np.int64), "sidesets": ss},
              dict(bc_sym=[1], bc_extrapolate=[2, 3]))]
    kw = dict(flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4, cfl=0.3)
    for g, bc in cases:
        ctx = capi.Context(4, options={"keep_connectivity": 1, "p1_rhs": 1}, **kw, **bc)
        m0 = capi.mesh_from_connectivity(ctx, g["inpoel"], g["coord"], g["sidesets"])
        m1 = m2 = m3 = None
        try:
            m0.state_initialize(0.0)
            t = 0.0
            for _ in range(2):
                t += m0.step(t)
            U0 = m0.state_download()
            m1, _ = m0.refine_uniform(host_copy=False)
            assert m1.nielem == 8 * m0.nielem
            m2 = m1.derefine_uniform("first_child")
            assert m2.nielem == m0.nielem
            assert np.array_equal(m2.state_download(), U0)                       # every DOF back
            assert np.array_equal(m2.lhs(), m0.lhs())
            assert np.array_equal(m2.rhs(t, U0), m0.rhs(t, U0)) and m2.dt(U0) == m0.dt(U0)
            for _ in range(2):
                assert m2.step(t) == m0.step(t)
            assert np.array_equal(m2.state_download(), m0.state_download())
            # it can be refined again (it keeps its connectivity), and that handle derefined again
            m3, _ = m2.refine_uniform(host_copy=False)
            assert m3.nielem == 8 * m0.nielem
            # conservation of the mean policy: evolve on the refined mesh, coarsen, compare the totals
            for _ in range(2):
                m1.step(t)
            U1 = m1.state_download().reshape(-1, 20)
            v1 = m1.lhs().reshape(-1, 20)[:, 0]                                  # L(e, 0) = vol_e (Mass.cpp:25-73)
            mc = m1.derefine_uniform("mean")
            try:
                Uc = mc.state_download().reshape(-1, 20)
                vc = mc.lhs().reshape(-1, 20)[:, 0]
                for c in (0, 4, 8, 12, 16):
                    a, b = (U1[:, c] * v1).sum(), (Uc[:, c] * vc).sum()
                    assert abs(a - b) <= 1e-13 * max(1.0, abs(a))
                assert np.abs(Uc[:, [1, 2, 3, 5, 6, 7]]).max() == 0.0
                assert abs(vc.sum() - v1.sum()) <= 1e-13 * v1.sum()
            finally:
                mc.close()
        finally:
            for m in (m0, m1, m2, m3):
                if m is not None:
                    m.close()
            ctx.close()
    # a mesh that is no uniform refinement is refused
    g = meshgen.kuhn_box(4, 2, 2)
    ctx = capi.Context(4, options={"keep_connectivity": 1}, **kw, bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])
    m = capi.mesh_from_connectivity(ctx, g["inpoel"], g["coord"], g["sidesets"])
    try:
        with pytest.raises(capi.QdgError):
            m.derefine_uniform()
    finally:
        m.close(); ctx.close()
