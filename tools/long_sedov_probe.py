import sys, json, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from conftest import load_fixture, compflow_err
from oracle import oracle as O
from quinoa_amd import capi, dgmesh
cases=json.load(open('tests/golden/cases.json'))
case, fix = cases["sedov_dgp1"], load_fixture("sedov_dgp1")
ss = {int(s): fix["ss_tri_%d" % s] for s in fix["ss_ids"]}
chunk = dgmesh.build_chunk(fix["coord"], fix["inpoel"], None, ss)
kw = dict(flux=case["flux"], limiter=case["limiter"], problem=case["problem"], gamma=case["gamma"])
for mode in (0,1):
    ctx = capi.Context(case["ndof"], cfl=case["cfl"], bc_dirichlet=case["bc_dirichlet"], bc_sym=case["bc_sym"], bc_extrapolate=case["bc_extrapolate"], options={"p1_rhs":mode}, **kw)
    mesh = dgmesh.upload(ctx, chunk)
    om = O.OracleMesh(fix["coord"], fix["inpoel"], ss)
    orc = O.Oracle(om, O.make_cfg(case["ndof"], **kw), case["bc_dirichlet"], case["bc_sym"], case["bc_extrapolate"])
    Lm = orc.lhs(); U = orc.initialize(Lm, 0.0); mesh.state_upload(U); t=0.0
    for step in range(150):
        dtg = mesh.step(t); dto = orc.step(t, U, Lm, cfl=case["cfl"]); t += dto
        if step in (0,4,19,49,99,149):
            Ug = mesh.state_download()
            print("p1_rhs", mode, "step", step+1, "dt rel", abs(dtg-dto)/dto, "global", np.abs(Ug-U).max()/max(1,np.abs(U).max()), "per-comp", compflow_err(Ug,U,4), flush=True)
    mesh.close(); ctx.close()
