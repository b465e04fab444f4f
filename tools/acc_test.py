# accuracy of the fast rcp/sqrt variants: R of the Sedov P1 case vs the oracle
import json, sys, numpy as np
sys.path.insert(0, '.')
from oracle import oracle as O
from quinoa_amd import capi, dgmesh
cases = json.load(open('tests/golden/cases.json'))
for name in ('sedov_dgp1', 'vortical_flow_dgp1'):
    case = cases[name]; fix = np.load('tests/golden/%s.npz' % name)
    ss = {int(s): fix["ss_tri_%d" % s] for s in fix["ss_ids"]}
    chunk = dgmesh.build_chunk(fix["coord"], fix["inpoel"], None, ss)
    kw = dict(flux=case["flux"], limiter=case["limiter"], problem=case["problem"], gamma=case["gamma"],
              alpha=case.get("alpha", 0.0), beta=case.get("beta", 0.0), p0=case.get("p0", 0.0))
    ctx = capi.Context(4, cfl=case["cfl"], dt=case["dt"], bc_dirichlet=case["bc_dirichlet"], bc_sym=case["bc_sym"], bc_extrapolate=case["bc_extrapolate"], **kw)
    mesh = dgmesh.upload(ctx, chunk)
    om = O.OracleMesh(fix["coord"], fix["inpoel"], ss)
    orc = O.Oracle(om, O.make_cfg(4, **kw), case["bc_dirichlet"], case["bc_sym"], case["bc_extrapolate"])
    Lm = orc.lhs(); U = orc.initialize(Lm, 0.0); t = 0
    for _ in range(3): t += orc.step(t, U, Lm, fixed_dt=case["dt"], cfl=case["cfl"])
    R = orc.rhs(t, U); Rg = mesh.rhs(t, U)
    print(name, "max |R_gpu - R_oracle| / max|R| = %.3e" % (np.abs(Rg - R).max() / np.abs(R).max()))
