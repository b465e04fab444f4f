"""BASELINE config 3: CompFlow vortical_flow, DG-P2 + WENO (wenop1, cweight 1),
alpha 0.1, beta 1, p0 10, gamma 5/3, Dirichlet on all six side sets
(SURVEY.md 8d cfg 3; reference: src/PDE/Limiter.cpp:29-153,
src/PDE/CompFlow/Problem/VorticalFlow.cpp:28-115).

The reference holds no DG-P2 + WENO regression baseline for CompFlow, so the
scheme is pinned by the oracle (itself pinned on the vortical_flow P0/P1 and
the DG-P2 TaylorGreen/NLEG goldens): every operator through the stateless
entry points and the resident time loop on the reference's `unitcube_1k`
fixture, then the size-independent properties at the config's full size
(110^3 x 6 = 7 986 000 tets).
"""
import numpy as np
import pytest

from conftest import compflow_err, load_fixture
from oracle import oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-10          # north_star: L_inf <= 1e-10 vs reference solution fields
CFG3 = dict(ndof=10, flux="hllc", limiter="wenop1", problem="vortical_flow",
            gamma=5.0 / 3.0, alpha=0.1, beta=1.0, p0=10.0, cweight=1.0)
SIDES = [1, 2, 3, 4, 5, 6]


def _pair(fix, dt):
    from quinoa_amd import capi, dgmesh
    ss = {int(s): fix["ss_tri_%d" % s] for s in fix["ss_ids"]}
    chunk = dgmesh.build_chunk(fix["coord"], fix["inpoel"], None, ss)
    ctx = capi.Context(CFG3["ndof"], flux=CFG3["flux"], limiter=CFG3["limiter"], problem=CFG3["problem"],
                       gamma=CFG3["gamma"], alpha=CFG3["alpha"], beta=CFG3["beta"], p0=CFG3["p0"],
                       cweight=CFG3["cweight"], dt=dt, bc_dirichlet=SIDES)
    mesh = dgmesh.upload(ctx, chunk)
    om = O.OracleMesh(fix["coord"], fix["inpoel"], ss)
    orc = O.Oracle(om, O.make_cfg(**CFG3), SIDES, [], [])
    return ctx, mesh, chunk, orc


def test_config3_operators_match_oracle():
    """lhs, initialize, rhs, dt, WENO limit of config 3's scheme, one stateless call each,
    on a state three oracle steps into the run (all ten modes populated)."""
    fix = load_fixture("vortical_flow_dgp1")          # the reference's unitcube_1k mesh
    dt = 1.0e-5
    ctx, mesh, chunk, orc = _pair(fix, dt)
    try:
        Lm = orc.lhs()
        assert np.abs(mesh.lhs() - Lm).max() <= 1e-15 * np.abs(Lm).max()
        U0 = orc.initialize(Lm, 0.0)
        assert np.abs(mesh.initialize(0.0) - U0).max() <= 1e-12 * max(1.0, np.abs(U0).max())
        U, t = U0.copy(), 0.0
        for _ in range(3):
            t += orc.step(t, U, Lm, fixed_dt=dt)
        R = orc.rhs(t, U)
        assert np.abs(mesh.rhs(t, U) - R).max() <= 1e-11 * max(1.0, np.abs(R).max())
        dto = orc.dt(U)
        assert abs(mesh.dt(U) - dto) <= 1e-12 * dto
        Ul = orc.limit(U.copy())
        assert np.abs(mesh.limit(U) - Ul).max() <= 1e-12 * max(1.0, np.abs(Ul).max())
        # WENO touches DOFs 1-3 only (Limiter.cpp:146-151), also at P2
        Ul2, U2 = Ul.reshape(-1, 5, 10), U.reshape(-1, 5, 10)
        assert np.array_equal(Ul2[:, :, 0], U2[:, :, 0]) and np.array_equal(Ul2[:, :, 4:], U2[:, :, 4:])
        assert np.abs(Ul2[:, :, 1:4] - U2[:, :, 1:4]).max() > 0.0
    finally:
        mesh.close(); ctx.close()


def test_config3_time_loop_matches_oracle():
    """resident run of config 3's scheme, 8 SSP-RK3 steps (3 x [WENO, RHS, update] each),
    every DOF vs the oracle; L2 diagnostics vs the oracle's"""
    fix = load_fixture("vortical_flow_dgp1")
    dt = 1.0e-5
    ctx, mesh, chunk, orc = _pair(fix, dt)
    try:
        mesh.state_initialize(0.0)
        Lm = orc.lhs(); U = orc.initialize(Lm, 0.0)
        t = 0.0
        for _ in range(8):
            dtg = mesh.step(t)
            assert dtg == dt
            orc.step(t, U, Lm, fixed_dt=dt)
            t += dt
        Ug = mesh.state_download()
        err = compflow_err(Ug, U, 10)
        assert err <= TOL, err
        d = mesh.diag(t)
        l2, _ = orc.diag(t, U)
        assert np.abs(np.sqrt(d[:10] / chunk.meshvol) - l2).max() <= 1e-10
        # WENO_P1 averages the reference-space DOFs 1-3 of differently shaped neighbours
        # (Limiter.cpp:88-144), so it perturbs even this smooth manufactured solution: the
        # oracle's own L2 errors after 8 steps are O(1e-2) -- compare, do not bound
        assert np.abs(np.sqrt(d[5:10] / chunk.meshvol) - orc.diag(t, U)[0][5:10]).max() <= 1e-10
    finally:
        mesh.close(); ctx.close()


def _submesh(chunk, sel):
    """tets `sel` of a chunk as a standalone mesh (nodes renumbered); returns coord, inpoel and
    the mask of its tets whose four face neighbours are all inside the sub-mesh"""
    inp = chunk.inpoel[sel]
    nodes, inv = np.unique(inp.reshape(-1), return_inverse=True)
    sub_inpoel = inv.reshape(-1, 4)
    inside = np.zeros(chunk.nunk, dtype=bool)
    inside[sel] = True
    nb = chunk.esuel.reshape(-1, 4)[sel]
    full = (nb >= 0).all(axis=1) & inside[np.maximum(nb, 0)].all(axis=1)
    return chunk.coord[nodes], sub_inpoel, full


def test_config3_full_size_properties():
    """Config 3 at its own size, 110^3 x 6 = 7 986 000 tets, DG-P2 + WENO:
    exact upload/download round trip, free-stream preservation (uniform state at rest,
    R == 0 to rounding), RHS and WENO limiter equal to the ORACLE on a 2k-tet corner
    sub-mesh cut out of the full mesh (tets whose four neighbours are in the cut), the
    stateless RHS equal to the resident stage path, finite and bounded state after full
    limited steps."""
    from quinoa_amd import capi, dgmesh, meshgen
    n = 110
    ch = meshgen.kuhn_box(n, n, n)
    chunk = dgmesh.build_chunk(ch["coord"], ch["inpoel"], None, ch["sidesets"])
    assert chunk.nielem == 7986000
    h = 1.0 / n
    dt = 1.0e-5 * h / 0.1          # cfg 3: dt 1e-5 on the ~0.1-sized fixture tets, h-scaled
    ctx = capi.Context(10, flux="hllc", limiter="wenop1", problem="vortical_flow", gamma=5.0 / 3.0,
                       alpha=0.1, beta=1.0, p0=10.0, cweight=1.0, dt=dt, bc_dirichlet=SIDES)
    mesh = dgmesh.upload(ctx, chunk)
    del ch
    try:
        ne = chunk.nunk
        # exact round trip through the renumbering (pattern unique per row and column)
        Uc = (np.arange(ne, dtype=np.float64)[:, None] * 64.0 + np.arange(50, dtype=np.float64)[None, :])
        mesh.state_upload(Uc.reshape(-1))
        back = mesh.state_download()
        assert np.array_equal(back, Uc.reshape(-1))
        del back
        # free stream: a uniform state at rest has zero flux divergence, so R is the
        # (state-independent) source integral of the manufactured solution alone
        # (VorticalFlow.cpp:80-115): two different uniform states give the same R on every
        # tet without a boundary face (Dirichlet faces see the vortical state).
        # Not to rounding at P2: the weights of the reference's 6-point triangle rule sum to
        # 1 + 7.45e-9 (Quadrature.cpp:300-339: 6 * (0.054975870996713638 + 0.1116907969117165)),
        # so surface and volume terms of a constant pressure cancel only to 7.45e-9 * dp * area
        # -- in the reference too.  Bound = that defect with the basis magnitude (<= 6) folded in.
        Uc[:] = 0.0; Uc[:, 0] = 1.3; Uc[:, 40] = 5.0
        R = mesh.rhs(0.0, Uc.reshape(-1)).reshape(ne, 50)
        Uc[:, 0] = 0.7; Uc[:, 40] = 11.0
        R -= mesh.rhs(0.0, Uc.reshape(-1)).reshape(ne, 50)
        interior = (chunk.esuel.reshape(-1, 4) >= 0).all(axis=1)
        dp = (11.0 - 5.0) * (5.0 / 3.0 - 1.0)
        assert np.abs(R[interior]).max() <= 7.45e-9 * dp * chunk.geoFace[0::7].max() * 6.0
        # mass and energy rows carry no pressure term: those cancel to rounding
        assert np.abs(R[interior][:, 0:10]).max() <= 1e-15 and np.abs(R[interior][:, 40:50]).max() <= 1e-15
        del R, Uc
        # a perturbed manufactured state (all ten modes populated, deterministic)
        U0 = mesh.initialize(0.0).reshape(ne, 50)
        rng = np.random.default_rng(5)
        U0 += 1e-3 * rng.standard_normal(U0.shape) * np.array([1.0] + [0.3] * 9)[None, :].repeat(5, 0).reshape(1, 50)
        Rs = mesh.rhs(0.3, U0.reshape(-1)).reshape(ne, 50)        # stateless
        Ul = mesh.limit(U0.reshape(-1)).reshape(ne, 50)
        # ... vs the oracle on a corner sub-mesh of the SAME mesh
        cen = chunk.geoElem.reshape(-1, 4)[:, 1:4]
        sel = np.nonzero((cen < 7.2 * h).all(axis=1))[0]
        scoord, sinpoel, full = _submesh(chunk, sel)
        assert full.sum() > 500
        om = O.OracleMesh(scoord, sinpoel, {})
        orc = O.Oracle(om, O.make_cfg(**CFG3), [], [], [])
        Usub = np.ascontiguousarray(U0[sel]).reshape(-1)
        Ro = orc.rhs(0.3, Usub).reshape(-1, 50)
        assert np.abs(Rs[sel][full] - Ro[full]).max() <= 1e-11 * max(1.0, np.abs(Ro[full]).max())
        Ulo = orc.limit(Usub.copy()).reshape(-1, 50)
        assert np.abs(Ul[sel][full] - Ulo[full]).max() <= 1e-12 * max(1.0, np.abs(Ulo).max())
        del Ul
        # resident stage path == stateless RHS: R = (U1 - U0) L / dt on the means (L = vol)
        mesh.state_upload(U0.reshape(-1))
        mesh.stage_rhs_dt(0, 0.3)
        mesh.stage_update(0)
        U1 = mesh.state_download().reshape(ne, 50)
        vol = chunk.geoElem[0::4]
        Rres = (U1[:, 0::10] - U0[:, 0::10]) * vol[:, None] / dt
        Rst = Rs[:, 0::10]
        # forming U1 = U0 + dt R / L and differencing again costs eps |U| vol / dt absolute
        bound = 8.0 * np.finfo(float).eps * np.abs(U0).max() * vol.max() / dt + 1e-12 * max(1.0, np.abs(Rst).max())
        assert np.abs(Rres - Rst).max() <= bound
        del Rs, U1, Rres, Rst
        # full limited SSP-RK3 steps from the unperturbed initial state
        mesh.state_initialize(0.0)
        Ui = mesh.state_download()
        t = 0.0
        for _ in range(2):
            t += mesh.step(t)
        U2 = mesh.state_download()
        assert np.isfinite(U2).all()
        assert abs(t - 2 * dt) <= 1e-15
        # WENO redistributes the reference-space slopes (see the time-loop test) but the means
        # move only through the RHS: two steps of dt change them by O(dt |R| / vol)
        dm = np.abs(U2.reshape(ne, 50)[:, 0::10] - Ui.reshape(ne, 50)[:, 0::10]).max()
        assert dm <= 1e-3
        d = mesh.diag(t)
        assert np.isfinite(d).all() and d[0] > 0.0
    finally:
        mesh.close(); ctx.close()
