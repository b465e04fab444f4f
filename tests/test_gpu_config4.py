"""BASELINE config 4 at ONE GPU's share: CompFlow Sedov blast wave, DG-P1 + Superbee, CFL 0.3,
110^3 x 6 = 7 986 000 tets (64 M tets over 8 GPUs; SURVEY.md 8d cfg 4: pressure 783.4112 in
x, y < 0.05, 1e-6 elsewhere, symmetry on the x-min / y-min (and z) faces, extrapolation on the
others).  The physics is pinned on the reference's own `sedov_blastwave_dgp1` baselines (single
chunk and 4 PEs) at fixture size (tests/test_gpu_parity.py, tests/test_gpu_partition.py); this
test covers what only the full size can show: the tile kernel over 32 k tiles, the fused CFL
reduction, exact state round trip, operators equal to the ORACLE on a sub-mesh cut out of the
blast corner of the same mesh, conservation over full limited steps.
"""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

CFG4 = dict(ndof=4, flux="hllc", limiter="superbeep1", problem="sedov_blastwave", gamma=1.4)


def _submesh(chunk, sel):
    inp = chunk.inpoel[sel]
    nodes, inv = np.unique(inp.reshape(-1), return_inverse=True)
    inside = np.zeros(chunk.nunk, dtype=bool)
    inside[sel] = True
    nb = chunk.esuel.reshape(-1, 4)[sel]
    full = (nb >= 0).all(axis=1) & inside[np.maximum(nb, 0)].all(axis=1)
    return chunk.coord[nodes], inv.reshape(-1, 4), full


def test_config4_chunk_full_size_properties():
    from quinoa_amd import capi, dgmesh, meshgen
    n = 110
    ch = meshgen.kuhn_box(n, n, n)
    chunk = dgmesh.build_chunk(ch["coord"], ch["inpoel"], None, ch["sidesets"])
    assert chunk.nielem == 7986000
    del ch
    ctx = capi.Context(4, flux="hllc", limiter="superbeep1", problem="sedov_blastwave", gamma=1.4, cfl=0.3,
                       bc_sym=[1, 3, 5, 6], bc_extrapolate=[2, 4])
    mesh = dgmesh.upload(ctx, chunk)
    try:
        ne, h = chunk.nunk, 1.0 / n
        # exact round trip through the renumbering
        Uc = np.arange(ne, dtype=np.float64)[:, None] * 32.0 + np.arange(20, dtype=np.float64)[None, :]
        mesh.state_upload(Uc.reshape(-1))
        assert np.array_equal(mesh.state_download(), Uc.reshape(-1))
        del Uc
        # the initial condition with every mode perturbed, operators vs the oracle on the blast
        # corner (the pressure jump of 8 orders of magnitude included)
        U0 = mesh.initialize(0.0).reshape(ne, 20)
        rng = np.random.default_rng(9)
        U0[:, 0] *= 1.0 + 1e-3 * rng.standard_normal(ne)
        for c in range(5):
            U0[:, 4 * c + 1:4 * c + 4] += 1e-4 * rng.standard_normal((ne, 3)) * np.abs(U0[:, 4 * c:4 * c + 1]).clip(1e-6)
        Rs = mesh.rhs(0.0, U0.reshape(-1)).reshape(ne, 20)
        Ul = mesh.limit(U0.reshape(-1)).reshape(ne, 20)
        cen = chunk.geoElem.reshape(-1, 4)[:, 1:4]
        sel = np.nonzero((cen[:, 0] < 9.2 * h) & (cen[:, 1] < 9.2 * h) & (cen[:, 2] < 5.2 * h))[0]
        scoord, sinpoel, full = _submesh(chunk, sel)
        assert full.sum() > 800
        om = O.OracleMesh(scoord, sinpoel, {})
        orc = O.Oracle(om, O.make_cfg(**CFG4), [], [], [])
        Usub = np.ascontiguousarray(U0[sel]).reshape(-1)
        Ro = orc.rhs(0.0, Usub).reshape(-1, 20)
        assert np.abs(Rs[sel][full] - Ro[full]).max() <= 1e-11 * max(1.0, np.abs(Ro[full]).max())
        Ulo = orc.limit(Usub.copy()).reshape(-1, 20)
        assert np.abs(Ul[sel][full] - Ulo[full]).max() <= 1e-12 * max(1.0, np.abs(Ulo).max())
        del Rs, Ul, U0
        # full limited SSP-RK3 steps with the CFL time step from the unperturbed state: finite,
        # positive density and pressure, mass and energy conserved while the blast has not
        # reached an extrapolation face (symmetry walls carry no mass or energy flux)
        mesh.state_initialize(0.0)
        vol = chunk.geoElem[0::4]
        Ui = mesh.state_download().reshape(ne, 20)
        m0, e0 = (Ui[:, 0] * vol).sum(), (Ui[:, 16] * vol).sum()
        t, dts = 0.0, []
        for _ in range(3):
            dt = mesh.step(t)
            assert dt > 0.0 and np.isfinite(dt)
            dts.append(dt); t += dt
        U3 = mesh.state_download().reshape(ne, 20)
        assert np.isfinite(U3).all() and U3[:, 0].min() > 0.0
        m3, e3 = (U3[:, 0] * vol).sum(), (U3[:, 16] * vol).sum()
        assert abs(m3 - m0) <= 1e-12 * m0 and abs(e3 - e0) <= 1e-11 * e0
        # the time step is the blast region's: min(vol / sum(area * (|vn| + a))) * cfl / 3 with
        # a = sqrt(1.4 * 783.4112 / 1) ~ 33 and vol / sum(area) ~ h / 30 for these tets
        assert 1e-3 * h / 33.0 < dts[0] < 5e-2 * h / 33.0
        d = mesh.diag(t)
        assert np.isfinite(d).all()
    finally:
        mesh.close(); ctx.close()
