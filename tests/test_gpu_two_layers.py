"""Two ghost layers (qdg_chunk_build_depth / meshgen.kuhn_box_chunk(depth=2), qdg_halo_set_depth): a rank limits
its layer-1 ghosts itself -- layer 2 completes their inputs (src/PDE/Limiter.cpp:29-316 read a tet's face
neighbours) -- and the exchange of the LIMITED solution (DG::lim -> comlim, src/Inciter/DG.cpp:1262-1282) is
dropped: 3 exchanges per SSP-RK3 step instead of 6.  The same limiter runs on the same inputs, so the result must
equal the one-layer run's and the single chunk's."""
import numpy as np
import pytest

from conftest import compflow_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _chunks(dims, parts, depth, general=None):
    from quinoa_amd import meshgen, partition
    n = parts[0] * parts[1] * parts[2] if general is None else general[1]
    if general is None:
        return [meshgen.kuhn_box_chunk(*dims, parts=parts, rank=r, depth=depth) for r in range(n)], None
    g = meshgen.kuhn_box(*dims)
    part = partition.partition(g["coord"], g["inpoel"], n, general[0])
    out = []
    for r in range(n):
        ch = partition.build_chunk(g["coord"], g["inpoel"], g["sidesets"], part, n, r, depth=depth)
        ch["gid"] = g["gid"][ch["gid"]]
        out.append(ch)
    return out, g


def _run_chunks(ctx, chunks, nstep, np_):
    from quinoa_amd import capi, dg
    meshes = [capi.mesh_from_connectivity(ctx, c["inpoel"], c["coord"], c["sidesets"], nielem=c["nielem"],
                                          elem_gid=c["gid"]) for c in chunks]
    try:
        for m in meshes:
            m.state_initialize(0.0)
        drv = dg.LocalChunks(ctx, meshes, chunks)
        t = 0.0
        for _ in range(nstep):
            t += drv.step(t)
        ntet = sum(c["nielem"] for c in chunks)
        U = np.zeros((ntet, np_))
        for c, m in zip(chunks, meshes):
            U[c["gid"][:c["nielem"]]] = m.state_download().reshape(-1, np_)[:c["nielem"]]
        return U, t, drv
    finally:
        for m in meshes:
            m.close()


@pytest.mark.parametrize("ndof,limiter,problem", [(4, "superbeep1", "sod_shocktube"), (10, "wenop1", "sod_shocktube"),
                                                  (4, "wenop1", "sedov_blastwave")])
@pytest.mark.parametrize("cut", ["2x2x2", "2x2x1", "rcb:5"])
def test_two_ghost_layers_equal_one_layer_and_the_single_chunk(ndof, limiter, problem, cut):
    """all chunks on this GPU (dg.LocalChunks, device-built meshes with global ids, qdg_halo_copy as the
    transport): depth 2 (3 exchanges per step) == depth 1 (6 exchanges) == single chunk, every tet and DOF;
    block cuts of the bench (up to 6 (rank, layer) neighbours incl. the edge-diagonal ranks) and a general RCB cut"""
    from quinoa_amd import capi, meshgen
    dims = (10, 8, 6)
    kw = dict(flux="hllc", limiter=limiter, problem=problem, gamma=1.4, cfl=0.3)
    kw.update(dict(bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2]) if problem == "sod_shocktube" else
              dict(bc_sym=[1, 3, 5, 6], bc_extrapolate=[2, 4]))
    if cut.startswith("rcb"):
        parts, general = None, ("rcb", int(cut.split(":")[1]))
    else:
        parts, general = tuple(int(v) for v in cut.split("x")), None
    np_ = 5 * ndof
    nstep = 3
    res = {}
    for depth in (1, 2):
        ctx = capi.Context(ndof, **kw)
        try:
            chunks, _ = _chunks(dims, parts, depth, general)
            if depth == 2:
                assert all(c["depth"] == 2 and c["nghost1"] > 0 and 2 in c["nbr_layer"] for c in chunks)
            U, t, drv = _run_chunks(ctx, chunks, nstep, np_)
            assert drv.deep == (depth == 2)
            res[depth] = (U, t)
        finally:
            ctx.close()
    one = meshgen.kuhn_box(*dims)
    ctx1 = capi.Context(ndof, **kw)
    # (faces oriented by the generator's global ids in all runs: where a face point has p <= 0 HLLC falls through
    # to the STORED right state, src/PDE/Integrate/Riemann/HLLC.hpp:93-124 -- DG-P2 across the Sod jump does)
    m1 = capi.mesh_from_connectivity(ctx1, one["inpoel"], one["coord"], one["sidesets"], elem_gid=one["gid"])
    try:
        m1.state_initialize(0.0)
        t1 = 0.0
        for _ in range(nstep):
            t1 += m1.step(t1)
        ref = np.zeros_like(res[1][0])
        ref[one["gid"]] = m1.state_download().reshape(-1, np_)
    finally:
        m1.close(); ctx1.close()
    for depth in (1, 2):
        U, t = res[depth]
        assert abs(t - t1) <= 1e-12 * t1, depth
        assert compflow_err(U.reshape(-1), ref.reshape(-1), ndof) <= TOL, depth
    # the two depths run the same kernels on the same inputs: equal far below the bar
    assert compflow_err(res[2][0].reshape(-1), res[1][0].reshape(-1), ndof) <= 1e-12


def test_two_ghost_layers_transport_scalars():
    """dg::Transport (three scalars, DG-P1 + Superbee) on 2 x 2 x 1 chunks with two ghost layers == single chunk"""
    from quinoa_amd import capi, meshgen
    parts, dims, ndof = (2, 2, 1), (8, 7, 5), 4
    kw = dict(pde="transport", flux="upwind", problem="slot_cyl", dt=2.0e-3, limiter="superbeep1", ncomp=3,
              bc_dirichlet=[1, 2, 3, 4], bc_extrapolate=[5, 6])
    np_ = 3 * ndof
    ctx = capi.Context(ndof, **kw)
    try:
        chunks, _ = _chunks(dims, parts, 2)
        U, t, drv = _run_chunks(ctx, chunks, 3, np_)
        assert drv.deep
    finally:
        ctx.close()
    one = meshgen.kuhn_box(*dims)
    ctx1 = capi.Context(ndof, **kw)
    m1 = capi.mesh_from_connectivity(ctx1, one["inpoel"], one["coord"], one["sidesets"])
    try:
        m1.state_initialize(0.0)
        for _ in range(3):
            m1.step(0.0)
        ref = np.zeros_like(U)
        ref[one["gid"]] = m1.state_download().reshape(-1, np_)
    finally:
        m1.close(); ctx1.close()
    assert np.abs(U - ref).max() <= TOL * max(1.0, np.abs(ref).max())


def test_set_depth_is_refused_where_it_cannot_work():
    from quinoa_amd import capi, dgmesh, meshgen
    ch = meshgen.kuhn_box_chunk(6, 5, 4, parts=(2, 1, 1), rank=0, depth=2)
    kw = dict(flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4, cfl=0.3,
              bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])
    ctx = capi.Context(4, **kw)
    ck = dgmesh.build_chunk(ch["coord"], ch["inpoel"], ch["nielem"], ch["sidesets"])
    host = dgmesh.upload(ctx, ck)                    # qdg_mesh_upload: esuel of the owned tets only
    dev = capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"], nielem=ch["nielem"])
    try:
        host.halo_setup(ch["nbr_rank"], ch["send_lists"], ch["recv_counts"])
        with pytest.raises(capi.QdgError):
            host.halo_set_depth(ch["nghost1"])
        with pytest.raises(capi.QdgError):
            dev.halo_set_depth(ch["nghost1"])        # before qdg_halo_setup
        dev.halo_setup(ch["nbr_rank"], ch["send_lists"], ch["recv_counts"])
        with pytest.raises(capi.QdgError):
            dev.halo_set_depth(len(ch["gid"]))       # more than the ghost rows
        dev.halo_set_depth(ch["nghost1"])
        dev.halo_set_depth(0)
    finally:
        host.close(); dev.close(); ctx.close()


@pytest.mark.parametrize("cut", ["2x2x1", "rcb:3"])
def test_device_remesh_of_chunks_with_two_ghost_layers(cut):
    """qdg_mesh_refine_chunk on handles with two ghost layers (keep_connectivity, halo_depth = 2): the refined
    chunk's layers and plan come from the rule of qdg_chunk_build_depth applied to the children around the old
    interface -- the same global ids, plan entries, send lists and receive counts as qdg_refine_chunk_depth on the
    host; the new handles keep limiting their layer-1 ghosts (3 exchanges per step) and the run equals the single
    chunk across the same refinement; a second re-mesh from the new handles works too."""
    from quinoa_amd import amr, capi, dg, meshgen
    dims = (6, 6, 4)
    kw = dict(flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4, cfl=0.3,
              bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])
    if cut.startswith("rcb"):
        parts, general = None, ("rcb", int(cut.split(":")[1]))
    else:
        parts, general = tuple(int(v) for v in cut.split("x")), None
    chunks, _ = _chunks(dims, parts, 2, general)
    ctx = capi.Context(4, options={"keep_connectivity": 1, "halo_depth": 2}, **kw)
    meshes = [capi.mesh_from_connectivity(ctx, c["inpoel"], c["coord"], c["sidesets"], nielem=c["nielem"],
                                          elem_gid=c["gid"]) for c in chunks]
    one = meshgen.kuhn_box(*dims)
    ctx1 = capi.Context(4, **kw)
    run = amr.RefinedRun(ctx1, one["coord"], one["inpoel"], one["sidesets"])
    try:
        for m in meshes:
            m.state_initialize(0.0)
        run.mesh.state_initialize(0.0)
        drv = dg.LocalChunks(ctx, meshes, chunks)
        assert drv.deep
        t = t1 = 0.0
        for _ in range(2):
            t += drv.step(t)
            t1 += run.mesh.step(t1)
        for level in range(2):
            new_meshes, new_chunks = [], []
            for c, m in zip(chunks, meshes):
                host, _par = amr.refine_chunk(c)
                m2, plan = m.refine_chunk()
                assert plan["depth"] == 2 and plan["nghost1"] == host["nghost1"] and plan["nielem"] == host["nielem"]
                assert plan["nbr_rank"] == host["nbr_rank"] and plan["nbr_layer"] == host["nbr_layer"]
                assert plan["recv_counts"] == host["recv_counts"]
                assert np.array_equal(plan["gid"], host["gid"])
                for a, b in zip(plan["send_lists"], host["send_lists"]):
                    assert np.array_equal(a, b)
                assert m2.halo_info()[1] == plan["nghost1"]
                m.close()
                # (the next level's host cross-check needs the refined chunk's mesh: take the host's)
                host["gid"] = plan["gid"]
                new_meshes.append(m2); new_chunks.append(host)
            meshes, chunks = new_meshes, new_chunks
            run.refine()
            drv = dg.LocalChunks(ctx, meshes, chunks)
            assert drv.deep
            for _ in range(2):
                t += drv.step(t)
                t1 += run.mesh.step(t1)
            assert abs(t - t1) <= 1e-12 * t1
            # single-chunk reference: child 8 * e + k of the generator's tet with global id gid(e): the chunks'
            # global child ids are 8 * gid + k, the single chunk numbers its children 8 * (local e) + k
            ref = run.mesh.state_download().reshape(-1, 20)
            gmap = one["gid"] if level == 0 else gmap_next
            gchild = (8 * np.asarray(gmap)[:, None] + np.arange(8)).reshape(-1)
            glob = np.zeros((len(gchild), 20))
            glob[gchild] = ref
            gmap_next = gchild
            for c, m in zip(chunks, meshes):
                nie = c["nielem"]
                U = m.state_download().reshape(-1, 20)[:nie]
                assert compflow_err(U.reshape(-1), glob[c["gid"][:nie]].reshape(-1), 4) <= TOL, level
    finally:
        for m in meshes:
            m.close()
        run.mesh.close(); ctx.close(); ctx1.close()
