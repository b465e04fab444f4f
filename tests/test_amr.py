"""Mesh refinement during time stepping (BASELINE config 5's loop): uniform 1:8 refinement
(qdg_refine_uniform) + the state hand-over of DG::resizePostAMR, pinned on the reference's own
DG regression case with t>0 refinement: tests/regression/inciter/mesh_refinement/dtref/
gauss_hump.q (dg::Transport, DG-P0, uniform refinement every 5 of 10 steps, 112 -> 896 -> 7 168
tets) and its committed goldens gauss_hump_u_trans_pe1_u0.0.std.e-s.{0,1,2}.1.0 + gauss_hump_dg.std.
Host code + CPU oracle only."""
import os

import numpy as np
import pytest

from conftest import load_fixture
from oracle import oracle as O
from quinoa_amd import amr


def _ss(fix):
    return {int(s): fix["ss_tri_%d" % s] for s in fix["ss_ids"]}


def test_uniform_refinement_reproduces_the_reference_refined_meshes():
    """the children of refine_one_to_eight (AMR/refinement.hpp:425-536): same tets in the same
    order and the same coordinates as the meshes inside the reference's golden files, twice"""
    fix = load_fixture("gauss_hump_dtref")
    c, i, s = fix["coord"], fix["inpoel"], _ss(fix)
    assert np.array_equal(fix["s0_inpoel"], i)
    for k in (1, 2):
        ne = i.shape[0]
        c, i, s, par = amr.refine_uniform(c, i, s)
        gc, gi = fix["s%d_coord" % k], fix["s%d_inpoel" % k]
        assert i.shape == gi.shape and c.shape == gc.shape
        assert np.array_equal(par, np.repeat(np.arange(ne), 8))
        # element by element: the same four points (node numbering of the new nodes may differ)
        assert np.abs(c[i].mean(axis=1) - gc[gi].mean(axis=1)).max() <= 1e-15
        # children tile the parent: positive volumes that add up
        v = np.einsum("ij,ij->i", c[i[:, 1]] - c[i[:, 0]], np.cross(c[i[:, 2]] - c[i[:, 0]], c[i[:, 3]] - c[i[:, 0]])) / 6.0
        assert v.min() > 0.0 and abs(v.sum() - 1.0) <= 1e-13
        # side sets: 4 children per boundary triangle, all still on the boundary of the new mesh
        om = O.OracleMesh(c, i, s)
        assert om.nbfac == sum(len(t) for t in s.values()) == int((om.esuel == -1).sum())


def test_refined_run_matches_reference_dtref_goldens(cases):
    """10 steps with uniform refinement after steps 5 and 10 (DG::refine, DG.cpp:1511-1534:
    It % dtfreq == 0), child <- parent (DG.cpp:1597-1605): every output time of the three golden
    files (c0_numerical, c0_analytic, c0_error) and the diagnostics table."""
    case, fix = cases["gauss_hump_dtref"], load_fixture("gauss_hump_dtref")
    c, i, s = fix["coord"], fix["inpoel"], _ss(fix)
    U, t, it, rows = None, 0.0, 0, []
    for k in range(3):
        f = {"coord": c, "inpoel": i, "ss_ids": np.array(sorted(s))}
        for sid in s:
            f["ss_tri_%d" % sid] = s[sid]
        nstep = case["dtfreq"] if k < 2 else 0
        r = O.run_transport_case(case, f, nstep=nstep, U0=U, t0=t, it0=it)
        gt, gv = fix["s%d_times" % k], fix["s%d_vals" % k]
        nout = len(gt)
        # the step that triggers the refinement is written on the NEW mesh: the old mesh's file
        # ends one output earlier
        assert np.allclose(r["times"][:nout], gt, rtol=0, atol=1e-15)
        assert np.abs(r["fields"][:nout] - gv[:, 0]).max() <= 1e-13          # c0_numerical
        m = r["mesh"]
        ge = m.geoElem.reshape(-1, 4)
        for j in range(nout):
            out = np.zeros((3, m.nelem))
            O.lib().orc_tr_field_output(O.TR_PROBLEM["gauss_hump"], O.C.c_int64(1), O.C.c_double(gt[j]),
                                        O.C.c_int64(m.nelem), O._p(m.geoElem, O.c_f64p),
                                        O._p(np.ascontiguousarray(r["fields"][j]), O.c_f64p), O._p(out, O.c_f64p))
            assert np.abs(out[1] - gv[j, 1]).max() <= 1e-14                   # c0_analytic
            assert np.abs(out[2] - gv[j, 2]).max() <= 1e-15                   # c0_error
        rows += [row for row in r["diag"]]
        U, t, it = r["U"], r["t"], it + nstep
        if k < 2:
            c, i, s, par = amr.refine_uniform(c, i, s)
            U = U.reshape(-1, 1)[par].reshape(-1)        # DG::resizePostAMR: child <- parent
    gold = {int(g[0]): g for g in fix["diag"]}
    assert sorted(int(r[0]) for r in rows) == sorted(gold)
    for r in rows:
        g = gold[int(r[0])]
        # L2(c0) and L2(c0 - analytic): the latter jumps by two orders after the first refinement
        # in the reference's table too (1.27e-3 -> 1.01e-1)
        assert abs(r[1] - g[1]) <= 1e-12 and abs(r[3] - g[3]) <= 6e-7 * g[3] and abs(r[4] - g[4]) <= 6e-7 * g[4], (r, g)


def test_partitioned_chunks_refine_independently_and_stay_consistent():
    """config 5 on several ranks: every rank refines ITS chunk (owned + ghost tets) alone
    (amr.refine_chunk); the refined chunks must be what the general chunk builder makes of the
    refined GLOBAL mesh with every child on its parent's rank -- same owned tets, same ghost layer,
    send lists that match the neighbours' ghost order -- and a partitioned oracle run across the
    refinement must equal the serial run on the refined mesh."""
    from quinoa_amd import dgmesh, meshgen, partition
    g = meshgen.kuhn_box(5, 4, 3)
    coord, inpoel, ss = g["coord"], g["inpoel"], g["sidesets"]
    nparts = 3
    part = partition.partition(coord, inpoel, nparts, "rcb")
    chunks = [partition.build_chunk(coord, inpoel, ss, part, nparts, r) for r in range(nparts)]
    new = [amr.refine_chunk(ch) for ch in chunks]
    # reference: refine the global mesh, children inherit the rank
    c2, i2, s2, par = amr.refine_uniform(coord, inpoel, ss)
    part2 = part[par]
    for r in range(nparts):
        ref = partition.build_chunk(c2, i2, s2, part2, nparts, r)
        ch, _ = new[r]
        nie = ch["nielem"]
        assert nie == ref["nielem"] and ch["nbr_rank"] == ref["nbr_rank"]
        assert np.array_equal(ch["gid"][:nie], ref["gid"][:nie])                     # owned children, same order
        assert np.array_equal(np.sort(ch["gid"][nie:]), np.sort(ref["gid"][nie:]))   # the same ghost layer
        assert ch["recv_counts"] == ref["recv_counts"]
        # tets carry the same coordinates
        a = ch["coord"][ch["inpoel"]].mean(axis=1); b = ref["coord"][ref["inpoel"]].mean(axis=1)
        oa, ob = np.argsort(ch["gid"], kind="stable"), np.argsort(ref["gid"], kind="stable")
        assert np.abs(a[oa] - b[ob]).max() <= 1e-15
        off = np.concatenate([[0], np.cumsum(ch["recv_counts"])])
        for i, q in enumerate(ch["nbr_rank"]):
            j = new[q][0]["nbr_rank"].index(r)
            sent = new[q][0]["gid"][new[q][0]["send_lists"][j]]
            assert np.array_equal(sent, ch["gid"][nie + off[i]:nie + off[i + 1]])    # q's send order = my ghost order
        ck = dgmesh.build_chunk(ch["coord"], ch["inpoel"], nie, ch["sidesets"])      # ghost faces all match
        assert ck.nfac > ck.nipfac


def test_partitioned_run_across_a_refinement_equals_serial_run():
    """config 5 end to end on 3 chunks with the oracle as the per-chunk numerics: 3 steps, every
    rank refines its own chunk and hands its state over (child <- parent), 3 more steps with the
    halo plan refine_chunk derived -- equal to the serial run across the same refinement."""
    import test_partition as TP
    from quinoa_amd import dgmesh, meshgen, partition
    g = meshgen.kuhn_box(5, 4, 3)
    kw = dict(flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4)
    bc = dict(bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])
    nparts, cfl = 3, 0.3

    def oracle_of(ch):
        ck = dgmesh.build_chunk(ch["coord"], ch["inpoel"], ch["nielem"], ch["sidesets"])
        cm = O.ChunkMesh(ck.coord, ck.inpoel, ck.nielem, ck.esuel, ck.esuf, ck.inpofa, ck.geoFace, ck.geoElem,
                         ck.bface, ck.nbfac)
        return O.Oracle(cm, O.make_cfg(4, **kw), **bc)

    def steps(runs, t, n):
        for _ in range(n):
            for stage in range(3):
                TP._exchange(runs, 20)
                for r in runs:
                    r["orc"].limit(r["U"])
                TP._exchange(runs, 20)
                if stage == 0:
                    dt = min(r["orc"].dt(r["U"]) for r in runs) * cfl / 3.0
                    for r in runs:
                        r["Un"] = r["U"].copy()
                for r in runs:
                    R = r["orc"].rhs(t, r["U"])
                    r["orc"].rk_update(stage, dt, r["Un"], R, r["L"], r["U"])
            t += dt
        return t

    part = partition.partition(g["coord"], g["inpoel"], nparts, "morton")
    runs = []
    for r in range(nparts):
        ch = partition.build_chunk(g["coord"], g["inpoel"], g["sidesets"], part, nparts, r)
        orc = oracle_of(ch)
        L = orc.lhs()
        runs.append({"ch": ch, "orc": orc, "L": L, "U": orc.initialize(L, 0.0)})
    t = steps(runs, 0.0, 3)
    for r in runs:
        ch2, par = amr.refine_chunk(r["ch"])
        U2 = r["U"].reshape(-1, 20)[par].reshape(-1)          # DG::resizePostAMR: child <- parent
        orc = oracle_of(ch2)
        r.update(ch=ch2, orc=orc, L=orc.lhs(), U=U2)
    t = steps(runs, t, 3)
    # serial
    om = O.OracleMesh(g["coord"], g["inpoel"], g["sidesets"])
    so = O.Oracle(om, O.make_cfg(4, **kw), **bc)
    L = so.lhs(); U = so.initialize(L, 0.0)
    ts = 0.0
    for _ in range(3):
        ts += so.step(ts, U, L, cfl=cfl)
    c2, i2, s2, par = amr.refine_uniform(g["coord"], g["inpoel"], g["sidesets"])
    U = U.reshape(-1, 20)[par].reshape(-1)
    so2 = O.Oracle(O.OracleMesh(c2, i2, s2), O.make_cfg(4, **kw), **bc)
    L2 = so2.lhs()
    for _ in range(3):
        ts += so2.step(ts, U, L2, cfl=cfl)
    assert abs(t - ts) <= 1e-12 * ts
    ref = U.reshape(-1, 20)
    # global id of a child = 8 * (global id of the parent) + k = its row in the serially refined mesh
    gmap = g["gid"]                                             # generator ids of the undivided mesh
    seen = 0
    for r in runs:
        nie = r["ch"]["nielem"]
        rows = r["ch"]["gid"][:nie]
        err = np.abs(r["U"].reshape(-1, 20)[:nie] - ref[rows]).max() / np.abs(ref).max()
        assert err <= 1e-12, err
        seen += nie
    assert seen == ref.shape[0] and gmap is not None


def test_refine_chunk_native_equals_numpy_statement():
    """qdg_refine_chunk (C++, the one the runs use) against the numpy statement of the same
    algorithm: every array of the refined chunk, its ghost layer and halo plan identical, for
    the chunks of a general 3-way cut and for a block chunk with three neighbours"""
    from quinoa_amd import amr, meshgen, partition
    g = meshgen.kuhn_box(6, 5, 4)
    part = partition.partition(g["coord"], g["inpoel"], 3, "rcb")
    chunks = [partition.build_chunk(g["coord"], g["inpoel"], g["sidesets"], part, 3, r) for r in range(3)]
    chunks.append(meshgen.kuhn_box_chunk(8, 6, 6, lengths=(1.0, 1.0, 1.0), parts=(2, 2, 2), rank=3))
    for ch in chunks:
        a, pa = amr.refine_chunk(ch)
        b, pb = amr._refine_chunk_numpy(ch)
        assert a["nielem"] == b["nielem"] and a["nbr_rank"] == b["nbr_rank"] and a["recv_counts"] == b["recv_counts"]
        for k in ("coord", "inpoel", "gid"):
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
        assert np.array_equal(pa, pb)
        for p, q in zip(a["send_lists"], b["send_lists"]):
            assert np.array_equal(p, q)
        assert sorted(a["sidesets"]) == sorted(b["sidesets"])
        for sid in a["sidesets"]:
            assert np.array_equal(a["sidesets"][sid], b["sidesets"][sid])


@pytest.mark.parametrize("parts", [(2, 1, 1), (2, 2, 2)])
def test_refinement_of_a_chunk_with_two_ghost_layers(parts):
    """qdg_refine_chunk_depth: every rank refines its two-layer chunk alone; the refined chunks' layers and plans
    (one entry per (rank, layer)) are those of qdg_chunk_build_depth on the refined undivided mesh with the
    children's owners -- same tets per entry, every pair's send list = the other side's receive range, ordered by
    global child id; parents point at the old chunk's tets."""
    from quinoa_amd import amr, meshgen
    NX, NY, NZ = 4, 4, 4
    world = parts[0] * parts[1] * parts[2]
    chunks = [meshgen.kuhn_box_chunk(NX, NY, NZ, parts=parts, rank=r, depth=2) for r in range(world)]
    new, pars = zip(*[amr.refine_chunk(ch) for ch in chunks])
    ntet = 6 * NX * NY * NZ
    owner = np.zeros(8 * ntet, dtype=np.int64) - 1
    for r, c in enumerate(new):
        assert c["depth"] == 2 and c["nielem"] == 8 * chunks[r]["nielem"]
        g = c["gid"][:c["nielem"]]
        assert (owner[g] == -1).all()
        owner[g] = r
        # child k of parent p: global id 8 * gid(p) + k, in order
        assert np.array_equal(g, (8 * chunks[r]["gid"][:chunks[r]["nielem"], None] + np.arange(8)).reshape(-1))
        assert np.array_equal(c["gid"] >> 3, chunks[r]["gid"][pars[r]])
    assert (owner >= 0).all()
    for r, c in enumerate(new):
        nie, n1 = c["nielem"], c["nghost1"]
        roff = np.concatenate([[0], np.cumsum(c["recv_counts"])])
        assert c["nbr_layer"] == sorted(c["nbr_layer"]) and roff[-1] == len(c["gid"]) - nie
        assert sum(n for n, l in zip(c["recv_counts"], c["nbr_layer"]) if l == 1) == n1
        for i, (q, l) in enumerate(zip(c["nbr_rank"], c["nbr_layer"])):
            seg = c["gid"][nie + roff[i]:nie + roff[i + 1]]
            assert (owner[seg] == q).all() and (np.diff(seg) > 0).all()
            o = new[q]
            j = [k for k, (qq, ll) in enumerate(zip(o["nbr_rank"], o["nbr_layer"])) if qq == r and ll == l]
            assert len(j) == 1
            assert np.array_equal(seg, o["gid"][o["send_lists"][j[0]]])
        # layer 1 = foreign children sharing a face with an owned child; layer 2 = foreign, not layer 1, sharing a
        # face with a layer-1 child: checked on the chunk's own connectivity (complete that far by construction)
        from quinoa_amd import capi
        es = capi.gen_esuel(c["inpoel"])
        lay = np.zeros(len(c["gid"]), dtype=int)
        lay[nie:nie + n1] = 1
        lay[nie + n1:] = 2
        for e in range(nie, nie + n1):
            assert any(nb >= 0 and nb < nie for nb in es[e])
        for e in range(nie + n1, len(c["gid"])):
            assert not any(nb >= 0 and nb < nie for nb in es[e])
            assert any(nb >= nie and lay[nb] == 1 for nb in es[e])


def _udu():
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "t0ref_gauss_hump_udu.npz"))
    st = []
    for k in range(6):
        ss = {int(s): f["s%d_ss_tri_%d" % (k, s)].astype(np.int64) for s in f["s%d_ss_ids" % k]}
        st.append((f["s%d_coord" % k], f["s%d_inpoel" % k].astype(np.int64), ss))
    return st


def _tri_keys(ss):
    return {s: set(map(tuple, np.sort(np.asarray(t), axis=1).tolist())) for s, t in ss.items()}


def test_uniform_derefinement_reproduces_the_reference_t0_sequence():
    """The reference's initial-refinement sequence uniform -> uniform_derefine -> uniform (-> again), the meshes its
    regression test amr_t0ref_ud(ud)u_trans_dg compares after every step
    (mesh_refinement/t0ref/gauss_hump_dg_uniform_deref_t0ref.std.e-s.{0..5}.1.0: 955 -> 7 640 -> 955 -> 7 640 ...
    tets; after the derefinement the reference holds the initial mesh again, array for array).  qdg_refine_uniform of
    stage 0 gives stage 1's tets; qdg_derefine_uniform of that gives stage 2 = stage 0 EXACTLY (connectivity,
    coordinates, side sets); refined again = stage 3; and once more.  Meshes that are no uniform refinement in the
    library's order are refused."""
    st = _udu()
    for a, b in ((2, 0), (4, 0), (3, 1), (5, 1)):                  # the reference's own data
        assert np.array_equal(st[a][0], st[b][0]) and np.array_equal(st[a][1], st[b][1])
    c, i, s = st[0]
    for cycle in range(2):
        c1, i1, s1, par = amr.refine_uniform(c, i, s)
        gc, gi, gs = st[2 * cycle + 1]
        assert i1.shape == gi.shape and c1.shape == gc.shape
        # the same tets (their four points) as the reference's refined mesh, and the same side-set surfaces
        key = lambda cc, ii: np.sort(np.round(cc[ii].mean(axis=1) * 1e12).astype(np.int64).view([("", np.int64)] * 3).reshape(-1))
        assert np.array_equal(key(c1, i1), key(gc, gi))
        assert {k: len(v) for k, v in s1.items()} == {k: len(v) for k, v in gs.items()}
        c2, i2, s2 = amr.derefine_uniform(c1, i1, s1)
        gc2, gi2, gs2 = st[2 * cycle + 2] if 2 * cycle + 2 < 6 else st[0]
        assert np.array_equal(i2, gi2) and np.array_equal(c2, gc2)          # the initial mesh again, exactly
        assert _tri_keys(s2) == _tri_keys(gs2)
        c, i, s = c2, i2, s2
    # the reference's OWN refined mesh (stage 1: its children and their local node order are this library's) goes
    # back to the reference's own derefined mesh (stage 2) through qdg_derefine_uniform, array for array
    c2, i2, _ = amr.derefine_uniform(st[1][0], st[1][1], {})
    assert np.array_equal(i2, st[2][1]) and np.array_equal(c2, st[2][0])
    # not a refinement in the library's order: refused
    with pytest.raises(Exception):
        amr.derefine_uniform(st[1][0], st[1][1][::-1].copy(), {})
    with pytest.raises(Exception):
        amr.derefine_uniform(st[0][0], st[0][1][:952], {})
