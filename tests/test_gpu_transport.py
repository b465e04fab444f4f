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
    ctx = capi.Context(ndof, pde="transport", flux="upwind", problem="slot_cyl", dt=case["dt"],
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
        fo, names = mesh.field_output()
        assert names == ["c0_numerical"] and np.array_equal(fo[0], U)
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
