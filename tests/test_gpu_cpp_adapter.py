"""The C++ DGPDE-shaped adapter (include/qdg_dgpde.hpp), driven like Inciter's DG
chare drives g_dgpde, against the oracle.  The driver is compiled by
__graft_entry__.build() with plain g++ against the C ABI only."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle as O

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "tests", "cpp", "test_dgpde_adapter")


def _read_vecs(path):
    out = []
    with open(path, "rb") as f:
        while True:
            h = f.read(8)
            if not h:
                break
            n = struct.unpack("<Q", h)[0]
            out.append(np.frombuffer(f.read(8 * n), dtype=np.float64))
    return out


def test_cpp_adapter_matches_oracle(tmp_path):
    from quinoa_amd import meshgen
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    ch = meshgen.kuhn_box(10, 6, 5)
    coord, inpoel = ch["coord"], ch["inpoel"]
    ids = sorted(ch["sidesets"])
    tri = np.concatenate([ch["sidesets"][s] for s in ids]).astype(np.uint64)
    tset = np.concatenate([np.full(len(ch["sidesets"][s]), s, np.int32) for s in ids])
    mesh = tmp_path / "mesh.bin"
    with open(mesh, "wb") as f:
        f.write(struct.pack("<QQQ", coord.shape[0], inpoel.shape[0], tri.shape[0]))
        for d in range(3):
            f.write(np.ascontiguousarray(coord[:, d]).tobytes())
        f.write(inpoel.astype(np.uint64).tobytes())
        f.write(tri.tobytes())
        f.write(tset.tobytes())
    out = tmp_path / "out.bin"
    r = subprocess.run([EXE, str(mesh), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    L, U, R, Ulim, U2, sc, Lt, Ut, Rt, Ut2, tsc, f_rho, f_p, asol, Us, Us2, ssol = _read_vecs(out)
    kw = dict(flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4)
    om = O.OracleMesh(coord, inpoel, ch["sidesets"])
    orc = O.Oracle(om, O.make_cfg(4, **kw), bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])
    Lo = orc.lhs(); Uo = orc.initialize(Lo, 0.0)
    assert np.abs(L - Lo).max() <= 1e-15 * Lo.max()
    assert np.abs(U - Uo).max() <= 1e-12
    Ro = orc.rhs(0.0, Uo)
    assert np.abs(R - Ro).max() <= 1e-11 * max(1.0, np.abs(Ro).max())
    assert abs(sc[0] - orc.dt(Uo)) <= 1e-12 * sc[0]
    assert np.abs(Ulim - orc.limit(Uo.copy())).max() <= 1e-12
    t = 0.0
    for s in range(2):
        dt = orc.step(t, Uo, Lo, cfl=0.3)
        assert abs(sc[1 + s] - dt) <= 1e-11 * dt
        t += dt
    assert np.abs(U2 - Uo).max() <= 1e-10
    assert sc[3] == 1.0      # the bad call threw qdg::Exception

    # TransportHIP (dg::Transport stand-in): slot_cyl DG-P0, all four BC kinds
    import sys
    tcase = dict(ndof=1, dt=5.0e-4, nstep=2, bc_dirichlet=[1, 2], bc_extrapolate=[3, 4],
                 bc_inlet=[5], bc_outlet=[6])
    tfix = {"coord": coord, "inpoel": inpoel, "ss_ids": np.array(ids)}
    for sid in ids:
        tfix["ss_tri_%d" % sid] = ch["sidesets"][sid]
    r0 = O.run_transport_case(tcase, tfix, nstep=0)
    assert np.abs(Lt - r0["L"]).max() <= 1e-15 * r0["L"].max()
    assert np.abs(Ut - r0["U"]).max() <= 1e-13
    assert tsc[0] == sys.float_info.max and tsc[1] == 5.0e-4 and tsc[2] == 5.0e-4
    r2 = O.run_transport_case(tcase, tfix, nstep=2)
    assert np.abs(Ut2 - r2["U"]).max() <= 1e-12
    assert np.isfinite(Rt).all() and np.abs(Rt).max() > 0.0

    # TransportHIP with two scalars, ShearDiff (DG-P1, t0 = 1): initial condition and two steps vs the oracle
    scase = dict(ndof=4, ncomp=2, problem="shear_diff", dt=2.0e-3, nstep=2, t0=1.0, u0=[0.7, -0.3],
                 **{"lambda": [0.4, 0.1, -0.2, 0.3]}, diffusivity=[3.0, 2.0, 1.0, 1.5, 2.5, 0.8],
                 bc_dirichlet=[1, 2, 3, 4, 5, 6], bc_extrapolate=[], bc_inlet=[], bc_outlet=[])
    s0 = O.run_transport_multi(scase, tfix, nstep=0)
    s2 = O.run_transport_multi(scase, tfix)
    assert np.abs(Us - s0["U"]).max() <= 1e-13 * max(1.0, np.abs(s0["U"]).max())
    assert np.abs(Us2 - s2["U"]).max() <= 1e-11 * max(1.0, np.abs(s2["U"]).max())
    assert ssol.shape == (2,) and (ssol > 0.0).all()

    # output-side members: fieldOutput on the final state, analyticSolution (Sod left state)
    fo = orc.field_output(U2)
    assert np.abs(f_rho - fo[0]).max() <= 1e-14 and np.abs(f_p - fo[5]).max() <= 1e-13
    assert np.allclose(asol, [1.0, 0.0, 0.0, 0.0, 1.0 / 0.4], rtol=1e-15, atol=0.0)


MODEL_EXE = os.path.join(ROOT, "tests", "cpp", "test_dgpde_model")


@pytest.mark.parametrize("problem,ndof", [("sod_shocktube", 4), ("taylor_green", 10)])
def test_adapter_is_a_dgpde_model_in_dg_setup_order(tmp_path, problem, ndof):
    """tests/cpp/test_dgpde_model.cpp restates inciter::DGPDE's Concept/Model type erasure
    (src/PDE/DGPDE.hpp:159-259), builds it from CompFlowHIP through the factory's
    std::function path and drives lhs -> initialize -> limiter -> dt -> rhs -> fieldNames /
    fieldOutput / avgElemToNode (DG::setup, lim, dt, solve, writeFields order) with no
    attach() call: every result vs the oracle."""
    from quinoa_amd import meshgen
    assert os.path.exists(MODEL_EXE), "run __graft_entry__.build() first"
    ch = meshgen.kuhn_box(7, 6, 5)
    coord, inpoel = ch["coord"], ch["inpoel"]
    ids = sorted(ch["sidesets"])
    tri = np.concatenate([ch["sidesets"][s] for s in ids]).astype(np.uint64)
    tset = np.concatenate([np.full(len(ch["sidesets"][s]), s, np.int32) for s in ids])
    mesh = tmp_path / "mesh.bin"
    with open(mesh, "wb") as f:
        f.write(struct.pack("<QQQ", coord.shape[0], inpoel.shape[0], tri.shape[0]))
        for d in range(3):
            f.write(np.ascontiguousarray(coord[:, d]).tobytes())
        f.write(inpoel.astype(np.uint64).tobytes())
        f.write(tri.tobytes())
        f.write(tset.tobytes())
    out = tmp_path / "out.bin"
    r = subprocess.run([MODEL_EXE, str(mesh), str(out), problem, str(ndof)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    vecs = _read_vecs(out)
    L, U, Ulim, R, sc = vecs[:5]
    nf = int(sc[1])
    fout = np.array(vecs[5:5 + nf])
    nodal = np.array(vecs[5 + nf:11 + nf])
    om = O.OracleMesh(coord, inpoel, ch["sidesets"])
    if problem == "taylor_green":
        cfg = O.make_cfg(ndof, flux="hllc", limiter="wenop1", problem=problem, gamma=5.0 / 3.0, cweight=10.0)
        orc = O.Oracle(om, cfg, [1, 2, 3, 4, 5, 6], [], [])
    else:
        cfg = O.make_cfg(ndof, flux="hllc", limiter="superbeep1", problem=problem, gamma=1.4)
        orc = O.Oracle(om, cfg, [], [3, 4, 5, 6], [1, 2])
    Lo = orc.lhs(); Uo = orc.initialize(Lo, 0.0)
    assert np.abs(L - Lo).max() <= 1e-15 * Lo.max()
    assert np.abs(U - Uo).max() <= 1e-12 * max(1.0, np.abs(Uo).max())      # initialize() with no mesh attached
    Ul = orc.limit(Uo.copy())
    assert np.abs(Ulim - Ul).max() <= 1e-12 * max(1.0, np.abs(Ul).max())   # limiter with no mesh attached
    Ro = orc.rhs(0.0, Ul)
    assert np.abs(R - Ro).max() <= 1e-11 * max(1.0, np.abs(Ro).max())
    assert abs(sc[0] - orc.dt(Ul)) <= 1e-12 * sc[0]
    fo = orc.field_output_all(Ul, 0.25)
    assert nf == fo.shape[0] == len(orc.field_names())
    fin = np.isfinite(fo)
    assert np.array_equal(np.isfinite(fout), fin)
    assert np.abs(fout[fin] - fo[fin]).max() <= 1e-12 * max(1.0, np.abs(fo[fin]).max())
    no = orc.avg_elem_to_node(Ul)
    assert np.abs(nodal - no).max() <= 1e-12 * max(1.0, np.abs(no).max())
    with open(str(out) + ".names") as fh:
        assert fh.read().split("\n")[:-1] == orc.field_names()      # DGPDE::fieldNames


LEVEL2_EXE = os.path.join(ROOT, "tests", "cpp", "test_level2_driver")


@pytest.mark.parametrize("nparts", [2, 3])
def test_level2_cpp_driver_partition_halo_steps_remesh(tmp_path, nparts, remesh="host"):
    """tests/cpp/test_level2_driver.cpp: integration Level 2 from C++ through the C ABI alone --
    qdg_partition -> qdg_chunk_build -> qdg_mesh_from_chunk -> qdg_halo_setup -> steps in the DG
    chare's stage order (dg.ci:57-70) with qdg_halo_pack / qdg_halo_copy / qdg_halo_unpack,
    qdg_stage_limit, qdg_stage_rhs_dt, the dt minimum, qdg_stage_update -> qdg_refine_chunk ->
    qdg_mesh_from_chunk -> qdg_state_transfer -> more steps.  No Python between the calls; the
    result is compared with the oracle's serial run across the same uniform refinement."""
    from quinoa_amd import amr, meshgen
    assert os.path.exists(LEVEL2_EXE), "run __graft_entry__.build() first"
    ch = meshgen.kuhn_box(6, 5, 4)
    coord, inpoel = ch["coord"], ch["inpoel"]
    ids = sorted(ch["sidesets"])
    tri = np.concatenate([ch["sidesets"][s] for s in ids]).astype(np.uint64)
    tset = np.concatenate([np.full(len(ch["sidesets"][s]), s, np.int32) for s in ids])
    mesh = tmp_path / "mesh.bin"
    with open(mesh, "wb") as f:
        f.write(struct.pack("<QQQ", coord.shape[0], inpoel.shape[0], tri.shape[0]))
        for d in range(3):
            f.write(np.ascontiguousarray(coord[:, d]).tobytes())
        f.write(inpoel.astype(np.uint64).tobytes())
        f.write(tri.tobytes())
        f.write(tset.tobytes())
    out = tmp_path / "out.bin"
    n0, n1 = 3, 3
    r = subprocess.run([LEVEL2_EXE, str(mesh), str(out), str(nparts), str(n0), str(n1)] +
                       ([remesh] if remesh in ("device", "deep") else []), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    raw = open(out, "rb").read()
    t_end, nd = struct.unpack_from("<dQ", raw, 0)
    off = 16
    dts = np.frombuffer(raw, np.float64, nd, off); off += 8 * nd
    (np_,) = struct.unpack_from("<Q", raw, off); off += 8
    assert np_ == nparts and nd == n0 + n1
    got = {}
    for _ in range(np_):
        (nie,) = struct.unpack_from("<Q", raw, off); off += 8
        gid = np.frombuffer(raw, np.uint64, nie, off); off += 8 * nie
        U = np.frombuffer(raw, np.float64, nie * 20, off).reshape(nie, 20); off += 8 * nie * 20
        for g, row in zip(gid, U):
            assert int(g) not in got                     # every tet is owned by exactly one chunk
            got[int(g)] = row
    # the oracle: serial run, uniform refinement (child 8 * parent + k: the chunks' global ids)
    kw = dict(flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4)
    bc = dict(bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])
    om = O.OracleMesh(coord, inpoel, ch["sidesets"])
    orc = O.Oracle(om, O.make_cfg(4, **kw), **bc)
    Lm = orc.lhs(); U = orc.initialize(Lm, 0.0)
    t = 0.0
    for s in range(n0):
        dt = orc.step(t, U, Lm, cfl=0.3)
        assert abs(dts[s] - dt) <= 1e-11 * dt
        t += dt
    c2, i2, s2, par = amr.refine_uniform(coord, inpoel, ch["sidesets"])
    U = U.reshape(om.nelem, -1)[par].reshape(-1)
    om2 = O.OracleMesh(c2, i2, s2)
    orc2 = O.Oracle(om2, O.make_cfg(4, **kw), **bc)
    L2 = orc2.lhs()
    for s in range(n1):
        dt = orc2.step(t, U, L2, cfl=0.3)
        assert abs(dts[n0 + s] - dt) <= 1e-11 * dt
        t += dt
    assert abs(t - t_end) <= 1e-12 * t
    U = U.reshape(om2.nelem, 20)
    assert sorted(got) == list(range(om2.nelem))
    G = np.array([got[g] for g in range(om2.nelem)])
    assert np.abs(G - U).max() <= 1e-10 * max(1.0, np.abs(U).max())


def test_level2_cpp_driver_with_two_ghost_layers(tmp_path):
    """the same C++ driver on chunks with TWO ghost layers (qdg_chunk_build_depth, qdg_halo_set_depth, one exchange
    per stage, qdg_refine_chunk_depth for the re-mesh): same comparison with the oracle"""
    test_level2_cpp_driver_partition_halo_steps_remesh(tmp_path, 3, remesh="deep")


def test_level2_cpp_driver_with_the_device_remesh(tmp_path):
    """the same C++ driver with its re-mesh done by qdg_mesh_refine_chunk (one call per rank, on the device) instead
    of qdg_refine_chunk + qdg_mesh_from_chunk + qdg_state_transfer: same comparison with the oracle"""
    test_level2_cpp_driver_partition_halo_steps_remesh(tmp_path, 3, remesh="device")
