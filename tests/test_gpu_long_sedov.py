"""Sedov DG-P1 + Superbee on the reference's fixture for 150 CFL steps (the reference's own baseline
stops after 20): GPU and oracle stay together step by step, i.e. the reference's quirks on this
path (HLLC's NaN fall-through at the 8-orders-of-magnitude pressure jump, the max rule of dt) are
the same in both over a long run, not only over the baseline's horizon.  Per-component tolerance 1e-12
(measured 4e-15 after 150 steps; the time steps agree to 1e-12)."""
import numpy as np
import pytest

from conftest import compflow_err, load_fixture
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_sedov_150_steps_gpu_tracks_oracle(cases):
    from quinoa_amd import capi, dgmesh
    case, fix = cases["sedov_dgp1"], load_fixture("sedov_dgp1")
    ss = {int(s): fix["ss_tri_%d" % s] for s in fix["ss_ids"]}
    chunk = dgmesh.build_chunk(fix["coord"], fix["inpoel"], None, ss)
    kw = dict(flux=case["flux"], limiter=case["limiter"], problem=case["problem"], gamma=case["gamma"])
    ctx = capi.Context(case["ndof"], cfl=case["cfl"], bc_dirichlet=case["bc_dirichlet"], bc_sym=case["bc_sym"],
                       bc_extrapolate=case["bc_extrapolate"], **kw)
    mesh = dgmesh.upload(ctx, chunk)
    om = O.OracleMesh(fix["coord"], fix["inpoel"], ss)
    orc = O.Oracle(om, O.make_cfg(case["ndof"], **kw), case["bc_dirichlet"], case["bc_sym"], case["bc_extrapolate"])
    try:
        Lm = orc.lhs(); U = orc.initialize(Lm, 0.0)
        mesh.state_upload(U)
        t = 0.0
        worst = 0.0
        for step in range(150):
            dtg = mesh.step(t)
            dto = orc.step(t, U, Lm, cfl=case["cfl"])
            assert abs(dtg - dto) <= 1e-12 * dto, step
            t += dto
            if step % 25 == 24 or step == 149:
                Ug = mesh.state_download()
                fin = np.isfinite(U)
                assert np.array_equal(fin, np.isfinite(Ug)), step
                worst = max(worst, compflow_err(Ug, U, case["ndof"]))       # per component
        # (measured, both DG-P1 kernels: 4e-15 after 150 steps, tools/long_sedov_probe.py)
        assert worst <= 1e-12
    finally:
        mesh.close(); ctx.close()
