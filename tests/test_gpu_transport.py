"""GPU parity of the scalar Transport path (BASELINE configs[0]: Inciter
Transport slot_cyl, DG-P0, Upwind, dt 5e-4, 5 steps, Dirichlet on side set 1,
on the reference's 31 304-tet fixture mesh), through the C ABI, against the
reference's golden diag table and 4-PE golden chunks and against the oracle;
plus DG-P1 / DG-P2 transport against the oracle."""
import numpy as np
import pytest

from conftest import load_fixture
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _gpu(case, fix, ndof):
    from quinoa_amd import capi, dgmesh
    ss = {int(s): fix["ss_tri_%d" % s] for s in fix["ss_ids"]}
    chunk = dgmesh.build_chunk(fix["coord"], fix["inpoel"], None, ss)
    ctx = capi.Context(ndof, pde="transport", flux="upwind", problem=case.get("problem", "slot_cyl"),
                       dt=case["dt"], limiter=case.get("limiter", "nolimiter"),
                       cweight=case.get("cweight", 1.0), pref=case.get("pref", False),
                       tolref=case.get("tolref", 0.1),
                       bc_dirichlet=case["bc_dirichlet"], bc_extrapolate=case["bc_extrapolate"],
                       bc_inlet=case["bc_inlet"], bc_outlet=case["bc_outlet"])
    return ctx, dgmesh.upload(ctx, chunk), chunk


def _centroid_order(c):
    q = np.round(c * 1e9).astype(np.int64)
    return np.lexsort((q[:, 2], q[:, 1], q[:, 0]))


def test_slot_cyl_config1_matches_reference_golden_and_oracle(cases):
    case, fix = cases["slot_cyl_dg"], load_fixture("slot_cyl_dg")
    ctx, mesh, chunk = _gpu(case, fix, 1)
    try:
        assert mesh.nprop == 1
        mesh.state_initialize(0.0)
        t, rows = 0.0, []
        for it in range(case["nstep"]):
            dt = mesh.step(t)
            assert dt == case["dt"]
            t += dt
            rows.append([it + 1, t, np.sqrt(mesh.diag(t)[0] / chunk.meshvol)])
        U = mesh.state_download()
        fo, names = mesh.field_output(t)
        # dg::Transport::fieldNames / fieldOutput (DGTransport.hpp:211-229, 248-279)
        assert names == ["c0_numerical", "c0_analytic", "c0_error"] and np.array_equal(fo[0], U)
        assert np.array_equal(fo[2], (fo[1] - fo[0]) ** 2 * chunk.geoElem[0::4])
    finally:
        mesh.close(); ctx.close()
    # reference diag table (iteration, time, dt, L2(c0)): 6 printed digits
    for row, g in zip(rows, fix["diag"]):
        assert int(row[0]) == int(g[0]) and abs(row[1] - g[1]) < 1e-12
        assert abs(row[2] - g[3]) <= 6e-6 * g[3], (row, g)
    # reference 4-PE golden chunks: cell values at the last output time, by centroid
    cent = chunk.geoElem.reshape(-1, 4)[:, 1:]
    oa, ob = _centroid_order(cent), _centroid_order(fix["chunk_centroid"])
    assert np.abs(cent[oa] - fix["chunk_centroid"][ob]).max() < 1e-12
    assert abs(t - float(fix["chunk_time_last"][0])) < 1e-14
    assert np.abs(U[oa] - fix["chunk_c0_last"][ob]).max() <= 1e-10     # north_star bar
    # oracle, full DOF vector
    r = O.run_transport_case(case, fix)
    assert np.abs(U - r["U"]).max() <= 1e-12
    assert abs(rows[-1][2] - r["diag"][-1][3]) <= 1e-12


@pytest.mark.parametrize("ndof", [4, 10])
def test_transport_p1_p2_match_oracle(cases, ndof):
    case = dict(cases["slot_cyl_dg"], ndof=ndof)
    fix = load_fixture("slot_cyl_dg")
    ctx, mesh, chunk = _gpu(case, fix, ndof)
    try:
        r0 = O.run_transport_case(case, fix, nstep=0)
        U0 = mesh.initialize(0.0)
        assert np.abs(U0 - r0["U"]).max() <= 1e-13
        Lm = mesh.lhs()
        assert np.abs(Lm.reshape(-1, ndof)[:, 0] - chunk.geoElem[0::4]).max() == 0.0
        mesh.state_upload(U0)
        t = 0.0
        for _ in range(2):
            t += mesh.step(t)
        U = mesh.state_download()
    finally:
        mesh.close(); ctx.close()
    r = O.run_transport_case(case, fix, nstep=2)
    assert np.abs(U - r["U"]).max() <= 1e-11 * max(1.0, np.abs(r["U"]).max())


def test_transport_config_errors():
    from quinoa_amd import capi
    with pytest.raises(capi.QdgError, match="constant dt"):
        capi.Context(1, pde="transport", flux="upwind", problem="slot_cyl", cfl=0.3)
    with pytest.raises(capi.QdgError, match="upwind"):
        capi.Context(1, pde="transport", flux="hllc", problem="slot_cyl", dt=1e-3)


MORE = ["cyl_advect_dg", "cyl_advect_dgp1", "cyl_advect_dgp1_weno", "gauss_hump_dgp1",
        "gauss_hump_dgp2", "gauss_hump_pdg",
        # round 4: GaussHump DG-P0, the cube case (diag table only) and the reference Refiner's t0-refined mesh
        "gauss_hump_dg", "gauss_hump_cube", "t0ref_gauss_hump_dg"]


@pytest.mark.parametrize("name", MORE)
def test_transport_regression_cases_match_reference_golden(name, cases):
    """CylAdvect / GaussHump (DG-P0/P1/P2, Superbee, WENO, p-adaptive) resident on
    the GPU vs the reference's golden cell values, ndof field and diag tables
    (L2, L2 error, Linf error), and the full DOF vector vs the oracle."""
    case, fix = cases[name], load_fixture(name)
    ctx, mesh, chunk = _gpu(case, fix, case["ndof"])
    try:
        mesh.state_initialize(0.0)
        t, rows = 0.0, []
        f0, names = mesh.field_output(0.0)
        fields, times, ndofs, allf = [f0[0]], [0.0], [mesh.ndofel_get()], [f0]
        for it in range(case["nstep"]):
            t += mesh.step(t)
            if (it + 1) % case["diag_interval"] == 0:
                d = mesh.diag(t)
                rows.append([it + 1, t, case["dt"], np.sqrt(d[0] / chunk.meshvol),
                             np.sqrt(d[5] / chunk.meshvol), d[10]])
            if (it + 1) % case["plot_interval"] == 0 or it + 1 == case["nstep"]:
                allf.append(mesh.field_output(t)[0])
                fields.append(allf[-1][0]); times.append(t); ndofs.append(mesh.ndofel_get())
        U = mesh.state_download()
    finally:
        mesh.close(); ctx.close()
    if "exo_vals" in fix:
        assert np.allclose(times, fix["exo_times"], rtol=1e-12, atol=1e-15)
        assert np.abs(np.array(fields) - fix["exo_vals"][:, 0]).max() <= 1e-10     # north_star bar
        if case.get("pref"):
            assert np.array_equal(np.array(ndofs), fix["exo_vals"][:, 1].astype(np.int64))
        # every element field of the golden file: c0_numerical, c0_analytic, c0_error (+ ndof)
        assert names == [str(n) for n in fix["exo_names_all"]]
        got, gold = np.array(allf), fix["exo_vals_all"]
        assert np.abs(got - gold).max() <= 1e-10
    for row, g in zip(rows, fix["diag"]):
        assert int(row[0]) == int(g[0])
        for a, b in zip(row[1:len(g)], g[1:]):
            assert abs(a - b) <= 6e-7 * abs(b) + 1e-13, (name, int(row[0]), a, b)
    r = O.run_transport_case(case, fix)
    # all DOFs; WENO's (1e-8 + |grad|)^-2 weights amplify rounding (measured 1.8e-11)
    assert np.abs(U - r["U"]).max() <= 1e-10 * max(1.0, np.abs(r["U"]).max())


def _gpu_multi(case, fix, ndof):
    from quinoa_amd import capi, dgmesh
    ss = {int(s): fix["ss_tri_%d" % s] for s in fix["ss_ids"]}
    chunk = dgmesh.build_chunk(fix["coord"], fix["inpoel"], None, ss)
    ctx = capi.Context(ndof, pde="transport", flux="upwind", problem=case["problem"], dt=case["dt"],
                       limiter=case.get("limiter", "nolimiter"), cweight=case.get("cweight", 1.0),
                       bc_dirichlet=case.get("bc_dirichlet", []), bc_extrapolate=case.get("bc_extrapolate", []),
                       bc_inlet=case.get("bc_inlet", []), bc_outlet=case.get("bc_outlet", []),
                       ncomp=case["ncomp"], u0=case.get("u0"), lam=case.get("lambda"),
                       diffusivity=case.get("diffusivity"))
    return ctx, dgmesh.upload(ctx, chunk), chunk


@pytest.mark.parametrize("ndof,limiter", [(1, "nolimiter"), (4, "superbeep1"), (4, "wenop1"), (10, "nolimiter")])
def test_multi_scalar_transport_matches_oracle(cases, ndof, limiter):
    """dg::Transport with three scalars (component::transport 3, DGTransport.hpp:84-85): slot_cyl gives
    scalar c the solution at T = t + 2 pi c / 3 (SlotCyl.cpp:45); rows are component-major.  State, RHS
    operator, diagnostics (5 slots per kind), field output (3 x ncomp fields) and Problem::solution vs
    the oracle's per-scalar runs; scalar 0 equals the single-scalar run of config 1."""
    fix = load_fixture("slot_cyl_dg")
    case = dict(cases["slot_cyl_dg"], ndof=ndof, ncomp=3, problem="slot_cyl", limiter=limiter, nstep=2)
    ctx, mesh, chunk = _gpu_multi(case, fix, ndof)
    try:
        assert mesh.nprop == 3 * ndof
        r0 = O.run_transport_multi(case, fix, nstep=0)
        U0 = mesh.initialize(0.0)
        assert np.abs(U0 - r0["U"]).max() <= 1e-13
        R = mesh.rhs(0.0, U0)
        mesh.state_upload(U0)
        t = 0.0
        for _ in range(case["nstep"]):
            t += mesh.step(t)
        U = mesh.state_download()
        d = mesh.diag(t)
        fo, names = mesh.field_output(t)
        pts = chunk.geoElem.reshape(-1, 4)[:50, 1:]
        sol = ctx.solution(pts, 0.3)
    finally:
        mesh.close(); ctx.close()
    r = O.run_transport_multi(case, fix)
    ne = chunk.geoElem.size // 4
    assert np.abs(U - r["U"]).max() <= 1e-10 * max(1.0, np.abs(r["U"]).max())
    # scalar 0 of the system = the single scalar of config 1
    single = O.run_transport_case(dict(case, ncomp=1), fix)
    assert np.abs(U.reshape(ne, 3, ndof)[:, 0] - single["U"].reshape(ne, ndof)).max() <= 1e-10
    assert np.isfinite(R).all() and np.abs(R).max() > 0.0
    # diagnostics: L2 / L2 error / Linf error per scalar in slots c, 5 + c, 10 + c
    for c in range(3):
        g = r["diag"][c][-1]
        assert abs(np.sqrt(d[c] / chunk.meshvol) - g[3]) <= 1e-11
        assert abs(np.sqrt(d[5 + c] / chunk.meshvol) - g[4]) <= 1e-11
        assert abs(d[10 + c] - g[5]) <= 1e-11
    assert d[3] == 0.0 and d[4] == 0.0
    # field output: numerical, analytic, error blocks of ncomp fields (DGTransport.hpp:211-228)
    assert names == ["c0_numerical", "c1_numerical", "c2_numerical", "c0_analytic", "c1_analytic",
                     "c2_analytic", "c0_error", "c1_error", "c2_error"]
    for c in range(3):
        assert np.array_equal(fo[c], U.reshape(ne, 3, ndof)[:, c, 0])
        assert np.array_equal(fo[6 + c], (fo[3 + c] - fo[c]) ** 2 * chunk.geoElem[0::4])
    # Problem::solution: scalar c at time t is scalar 0 at t + 2 pi c / 3
    assert sol.shape == (50, 3)


@pytest.mark.parametrize("ndof", [4, 10])
def test_shear_diff_transport_matches_oracle(ndof):
    """TransportProblemShearDiff (ShearDiff.cpp:28-160) with two scalars of different u0 / lambda /
    diffusivity, started at t0 = 1 (the solution is singular at 0), Dirichlet on the whole boundary: no
    DG regression case of the reference runs it (its tests use the CG scheme), so the oracle is the
    only pin -- initial condition, analytic state and time stepping."""
    from quinoa_amd import meshgen
    ch = meshgen.kuhn_box(5, 4, 4)
    fix = {"coord": ch["coord"], "inpoel": ch["inpoel"], "ss_ids": np.array(sorted(ch["sidesets"]))}
    for s_, tri in ch["sidesets"].items():
        fix["ss_tri_%d" % s_] = tri
    case = {"ndof": ndof, "ncomp": 2, "problem": "shear_diff", "dt": 2.0e-3, "nstep": 3, "t0": 1.0,
            "u0": [0.7, -0.3], "lambda": [0.4, 0.1, -0.2, 0.3], "diffusivity": [3.0, 2.0, 1.0, 1.5, 2.5, 0.8],
            "bc_dirichlet": [int(s_) for s_ in sorted(ch["sidesets"])], "bc_extrapolate": [], "bc_inlet": [],
            "bc_outlet": [], "diag_interval": 1}
    ctx, mesh, chunk = _gpu_multi(case, fix, ndof)
    try:
        U0 = mesh.initialize(case["t0"])
        r0 = O.run_transport_multi(case, fix, nstep=0)
        assert np.abs(U0 - r0["U"]).max() <= 1e-13 * max(1.0, np.abs(r0["U"]).max())
        mesh.state_upload(U0)
        t = case["t0"]
        for _ in range(case["nstep"]):
            t += mesh.step(t)
        U = mesh.state_download()
        d = mesh.diag(t)
        pts = chunk.geoElem.reshape(-1, 4)[:20, 1:]
        sol = ctx.solution(pts, 1.5)
    finally:
        mesh.close(); ctx.close()
    r = O.run_transport_multi(case, fix)
    assert abs(t - r["t"]) < 1e-13
    assert np.abs(U - r["U"]).max() <= 1e-11 * max(1.0, np.abs(r["U"]).max())
    for c in range(2):
        g = r["diag"][c][-1]
        assert abs(np.sqrt(d[c] / chunk.meshvol) - g[3]) <= 1e-12 * max(1.0, g[3])
        assert abs(d[10 + c] - g[5]) <= 1e-12
    # the analytic solution at a few points, straight from ShearDiff.cpp:47-66
    for c in range(2):
        l0, l1 = case["lambda"][2 * c:2 * c + 2]; d0, d1, d2 = case["diffusivity"][3 * c:3 * c + 3]
        tt, x, y, z = 1.5, pts[:, 0], pts[:, 1], pts[:, 2]
        phi3s = (l0 * l0 * d1 / d0 + l1 * l1 * d2 / d0) / 12.0
        ref = 1.0 / (8.0 * np.pi ** 1.5 * np.sqrt(d0 * d1 * d2) * tt ** 1.5 * np.sqrt(1.0 + phi3s * tt * tt)) * \
            np.exp(-(x - case["u0"][c] * tt - 0.5 * (l0 * y + l1 * z) * tt) ** 2 / (4.0 * d0 * tt * (1.0 + phi3s * tt * tt))
                   - y * y / (4.0 * d1 * tt) - z * z / (4.0 * d2 * tt))
        assert np.abs(sol[:, c] - ref).max() <= 1e-14 * max(1.0, np.abs(ref).max())


def test_multi_scalar_config_errors():
    from quinoa_amd import capi
    with pytest.raises(capi.QdgError, match="ncomp"):
        capi.Context(1, pde="transport", flux="upwind", problem="slot_cyl", dt=1e-3, ncomp=6)
    with pytest.raises(capi.QdgError, match="shear_diff needs"):
        capi.Context(4, pde="transport", flux="upwind", problem="shear_diff", dt=1e-3, ncomp=2)
    with pytest.raises(capi.QdgError, match="one transported scalar"):
        capi.Context(4, pde="transport", flux="upwind", problem="slot_cyl", dt=1e-3, ncomp=2, pref=True)
