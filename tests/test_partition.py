"""General mesh decomposition (qdg_partition / qdg_chunk_build) on the reference's own
fixtures: structural properties of the cut and of the ghost / halo plan, and a partitioned
run -- per-chunk numerics by the CPU oracle, halo rows moved by plain copies in the stage
order of the DG chare -- against the serial run and against the reference's committed
4-PE goldens (sedov_blastwave_dgp1_pe4.std.exo.{0-3}, sedov_blastwave_pdg_pe4_u0.0...),
matched by tet centroid.  No GPU needed: libqdg's host mirrors only.
"""
import numpy as np
import pytest

from conftest import load_fixture
from oracle import oracle as O
from quinoa_amd import dgmesh, partition


TOL = 1e-12


def _centroid_order(c):
    q = np.round(np.asarray(c) * 1e9).astype(np.int64)
    return np.lexsort((q[:, 2], q[:, 1], q[:, 0]))


def _sidesets(fix):
    return {int(s): fix["ss_tri_%d" % s] for s in fix["ss_ids"]}


@pytest.mark.parametrize("method", ["rcb", "morton"])
@pytest.mark.parametrize("nparts", [1, 2, 3, 4, 8])
def test_partition_and_ghost_plan_properties(method, nparts):
    fix = load_fixture("sedov_dgp1")                      # unitsquare_01_3.6k.exo, 3 643 tets
    coord, inpoel = fix["coord"], fix["inpoel"]
    ne = inpoel.shape[0]
    part = partition.partition(coord, inpoel, nparts, method)
    assert part.min() == 0 and part.max() == nparts - 1
    sizes = np.bincount(part, minlength=nparts)
    assert sizes.max() - sizes.min() <= max(1, nparts // 2)          # balanced cut
    assert np.array_equal(part, partition.partition(coord, inpoel, nparts, method))   # deterministic
    om = O.OracleMesh(coord, inpoel, _sidesets(fix))                  # esuel of the whole mesh
    esuel = om.esuel.reshape(-1, 4)
    chunks = [partition.build_chunk(coord, inpoel, _sidesets(fix), part, nparts, r) for r in range(nparts)]
    seen = np.zeros(ne, dtype=int)
    for r, ch in enumerate(chunks):
        nie, gid = ch["nielem"], ch["gid"]
        own = gid[:nie]
        assert np.array_equal(own, np.nonzero(part == r)[0])         # owned tets, input order
        seen[own] += 1
        # ghosts = exactly the face neighbours of owned tets that live elsewhere (DG.cpp:468-712)
        nb = esuel[own].reshape(-1)
        want = np.unique(nb[(nb >= 0) & (part[np.maximum(nb, 0)] != r)])
        assert np.array_equal(np.sort(gid[nie:]), want)
        # grouped by owner rank, ascending rank, ascending global id inside a group
        off = np.concatenate([[0], np.cumsum(ch["recv_counts"])])
        assert ch["nbr_rank"] == sorted(ch["nbr_rank"]) and off[-1] == len(gid) - nie
        for i, q in enumerate(ch["nbr_rank"]):
            g = gid[nie + off[i]:nie + off[i + 1]]
            assert (part[g] == q).all() and (np.diff(g) > 0).all()
            # what q sends to r is what r stores as ghosts of q, in that order
            j = chunks[q]["nbr_rank"].index(r)
            assert np.array_equal(chunks[q]["gid"][chunks[q]["send_lists"][j]], g)
        # local connectivity is the global one renumbered; coordinates follow
        assert np.array_equal(ch["node_gid"][ch["inpoel"]], inpoel[gid])
        assert np.array_equal(ch["coord"], coord[ch["node_gid"]])
        # every boundary face of an owned tet is found again through the restricted side sets
        ck = dgmesh.build_chunk(ch["coord"], ch["inpoel"], nie, ch["sidesets"])
        nfree_owned = int((esuel[own] == -1).sum())
        assert ck.nbfac == nfree_owned
    assert (seen == 1).all()


def _chunk_oracles(fix, case, nparts, method):
    coord, inpoel, ss = fix["coord"], fix["inpoel"], _sidesets(fix)
    part = partition.partition(coord, inpoel, nparts, method)
    runs = []
    for r in range(nparts):
        ch = partition.build_chunk(coord, inpoel, ss, part, nparts, r)
        ck = dgmesh.build_chunk(ch["coord"], ch["inpoel"], ch["nielem"], ch["sidesets"])
        cm = O.ChunkMesh(ck.coord, ck.inpoel, ck.nielem, ck.esuel, ck.esuf, ck.inpofa, ck.geoFace, ck.geoElem,
                         ck.bface, ck.nbfac)
        cfg = O.make_cfg(case["ndof"], flux=case["flux"], limiter=case["limiter"], problem=case["problem"],
                         gamma=case["gamma"])
        orc = O.Oracle(cm, cfg, case["bc_dirichlet"], case["bc_sym"], case["bc_extrapolate"],
                       pref=case.get("pref", False), tolref=case.get("tolref", 0.1))
        runs.append({"ch": ch, "ck": ck, "orc": orc})
    return runs


def _exchange(runs, nprop, with_ndof=False):
    """comsol / comlim: every chunk's ghost rows <- the owners' rows (DG.cpp:1023-1086, 1262-1282);
    with p-adaptive DG the tets' ndof travels along (DG.cpp:1032, 1275)"""
    for rank, me in enumerate(runs):
        ch, nie = me["ch"], me["ch"]["nielem"]
        Um = me["U"].reshape(-1, nprop)
        off = np.concatenate([[0], np.cumsum(ch["recv_counts"])])
        for i, q in enumerate(ch["nbr_rank"]):
            src = runs[q]
            j = src["ch"]["nbr_rank"].index(rank)
            rows = src["ch"]["send_lists"][j]
            Um[nie + off[i]:nie + off[i + 1]] = src["U"].reshape(-1, nprop)[rows]
            if with_ndof:
                me["orc"].ndofel[nie + off[i]:nie + off[i + 1]] = src["orc"].ndofel[rows]


def _partitioned_oracle_run(fix, case, nparts, method, nstep=None):
    """the DG chare's stage loop over all chunks of a cut; with scheme pdg also eval_ndof
    (DG::next), ndof piggy-backed on both exchanges, propagate_ndof (DG::lim) and the zeroing of
    P0 tets (DG::solve) at stage 0"""
    runs = _chunk_oracles(fix, case, nparts, method)
    nprop = 5 * case["ndof"]
    pdg = bool(case.get("pref"))
    for r in runs:
        r["L"] = r["orc"].lhs()
        r["U"] = r["orc"].initialize(r["L"], 0.0)
    t = 0.0
    for _ in range(nstep or case["nstep"]):
        for stage in range(3):
            if pdg and stage == 0:
                for r in runs:
                    r["orc"].eval_ndof(r["U"])                   # DG::next
            _exchange(runs, nprop, pdg)                          # DG::next -> comsol
            if pdg and stage == 0:
                for r in runs:
                    r["orc"].propagate_ndof()                    # DG::lim
            for r in runs:
                r["orc"].limit(r["U"])                           # DG::lim
            _exchange(runs, nprop, pdg)                          # -> comlim
            if stage == 0:
                p = {1: 0.0, 4: 1.0, 10: 2.0}[case["ndof"]]
                dt = min(r["orc"].dt(r["U"]) for r in runs) * case["cfl"] / (2.0 * p + 1.0)   # contribute(min)
                for r in runs:
                    if pdg:
                        r["orc"].pdg_zero(r["U"])                # DG::solve
                    r["Un"] = r["U"].copy()
            for r in runs:
                R = r["orc"].rhs(t, r["U"])
                r["orc"].rk_update(stage, dt, r["Un"], R, r["L"], r["U"])
        t += dt
    ne = fix["inpoel"].shape[0]
    U = np.zeros((ne, nprop))
    for r in runs:
        nie = r["ch"]["nielem"]
        U[r["ch"]["gid"][:nie]] = r["U"].reshape(-1, nprop)[:nie]
    return U, t, runs


@pytest.mark.parametrize("method,nparts", [("rcb", 4), ("morton", 4), ("rcb", 3)])
def test_partitioned_sedov_dgp1_matches_serial_and_reference_pe4_goldens(method, nparts, cases):
    """config 4's physics on a partitioned mesh: Sedov DG-P1 + Superbee, CFL 0.3, 20 steps on the
    reference's unitsquare_01_3.6k in 3 / 4 chunks (both cut methods).  Against (a) the serial run
    and (b) the reference's own 4-PE baseline sedov_blastwave_dgp1_pe4.std.exo.{0-3}, tets matched
    by centroid.  The reference's harness accepts relative 1e-7 (exodiff_dg.cfg); measured here:
    5e-16 vs the serial run, 6e-16 vs the 4-PE golden -- only the order of a tet's face sums
    differs between cuts -- so the bound is 1e-12."""
    case, fix = cases["sedov_dgp1"], load_fixture("sedov_dgp1")
    U, t, _ = _partitioned_oracle_run(fix, case, nparts, method)
    om = O.OracleMesh(fix["coord"], fix["inpoel"], _sidesets(fix))
    orc = O.Oracle(om, O.make_cfg(4, flux=case["flux"], limiter=case["limiter"], problem=case["problem"],
                                  gamma=case["gamma"]), case["bc_dirichlet"], case["bc_sym"], case["bc_extrapolate"])
    Lm = orc.lhs(); Us = orc.initialize(Lm, 0.0)
    ts = 0.0
    for _ in range(case["nstep"]):
        ts += orc.step(ts, Us, Lm, cfl=case["cfl"])
    assert abs(t - ts) <= 1e-12 * ts
    got, ser = orc.field_output(U.reshape(-1)), orc.field_output(Us)
    scale = np.maximum(1.0, np.abs(ser).max(axis=1))[:, None]
    assert (np.abs(got - ser) / scale).max() <= TOL
    # the reference's partitioned baseline
    assert abs(t - float(fix["chunk_time_last"][0])) <= 1e-12 * t
    cent = om.geoElem.reshape(-1, 4)[:, 1:]
    oa, ob = _centroid_order(cent), _centroid_order(fix["chunk_centroid"])
    assert np.abs(cent[oa] - fix["chunk_centroid"][ob]).max() < 1e-12
    gold = fix["chunk_vals_last"][:, ob]
    assert (np.abs(got[:, oa] - gold) / scale).max() <= TOL


def test_partitioned_sedov_pdg_matches_reference_pe4_goldens(cases):
    """p-adaptive DG across chunk boundaries (ndof piggy-backed on comsol / comlim, DG.cpp:1032,
    1275): 4 chunks vs the reference's sedov_blastwave_pdg_pe4_u0.0.std.exo.{0-3} (4 chares) and
    sedov_blastwave_pdg_pe4_u0.9.std.exo.{0-39} (40 chares, over-decomposed + migrated) --
    solution fields and the per-element ndof field, which must be IDENTICAL -- and vs the serial
    run (measured 4e-16)."""
    case, fix = cases["sedov_pdg"], load_fixture("sedov_pdg")
    U, t, runs = _partitioned_oracle_run(fix, case, 4, "rcb")
    ne = fix["inpoel"].shape[0]
    ndof = np.zeros(ne, dtype=np.int64)
    for r in runs:
        nie = r["ch"]["nielem"]
        ndof[r["ch"]["gid"][:nie]] = r["orc"].ndofel[:nie]
    r1 = O.run_case(case, fix)
    ser = r1["fields"][-1]
    om = r1["mesh"]
    got = r1["oracle"].field_output(U.reshape(-1))
    scale = np.maximum(1.0, np.abs(ser).max(axis=1))[:, None]
    assert abs(t - r1["t"]) <= 1e-12 * t
    assert (np.abs(got - ser) / scale).max() <= TOL
    assert np.array_equal(ndof, r1["ndof"][-1])              # the same tets are P1
    cent = om.geoElem.reshape(-1, 4)[:, 1:]
    for tag in ("chunk", "ochunk"):                          # 4 chares, and 40 chares (-u 0.9)
        oa, ob = _centroid_order(cent), _centroid_order(fix[tag + "_centroid"])
        assert np.abs(cent[oa] - fix[tag + "_centroid"][ob]).max() < 1e-12
        gold = fix[tag + "_vals_last"][:, ob]
        assert abs(t - float(fix[tag + "_time_last"][0])) <= 1e-12 * t
        assert (np.abs(got[:, oa] - gold[:6]) / scale).max() <= TOL
        assert np.array_equal(ndof[oa], gold[6].astype(np.int64))


@pytest.mark.parametrize("parts", [(2, 1, 1), (2, 2, 1), (2, 2, 2), (3, 2, 1)])
def test_two_ghost_layers_of_the_two_chunk_builders_agree(parts):
    """Chunks with TWO ghost layers (the rank limits its layer-1 ghosts itself, 3 exchanges per step): the
    analytic block cut (meshgen.kuhn_box_chunk(depth=2)) and the general builder (qdg_chunk_build_depth on the
    undivided mesh with the same owners) produce the same layers and the same plan -- one entry per
    (neighbour rank, layer), layer-1 entries first; every pair's send list is the other side's receive range,
    tet by tet; the layer-1 prefix is the depth-1 chunk; layer 2 is exactly the set of foreign tets at face
    distance two (checked against a breadth-first search on the undivided mesh)."""
    from quinoa_amd import capi, meshgen, partition
    NX, NY, NZ = 6, 5, 4
    g = meshgen.kuhn_box(NX, NY, NZ)
    world = parts[0] * parts[1] * parts[2]
    chunks = [meshgen.kuhn_box_chunk(NX, NY, NZ, parts=parts, rank=r, depth=2) for r in range(world)]
    one = [meshgen.kuhn_box_chunk(NX, NY, NZ, parts=parts, rank=r, depth=1) for r in range(world)]
    owner = np.zeros(NX * NY * NZ * 6, dtype=np.int32)
    for r, ch in enumerate(chunks):
        owner[ch["gid"][:ch["nielem"]]] = r
    part_g = owner[g["gid"]]
    esuel = capi.gen_esuel(g["inpoel"])
    gens = []
    for r, ch in enumerate(chunks):
        gen = partition.build_chunk(g["coord"], g["inpoel"], g["sidesets"], part_g, world, r, depth=2)
        gen["gid"] = g["gid"][gen["gid"]]
        gens.append(gen)
        nie, n1 = ch["nielem"], ch["nghost1"]
        assert ch["depth"] == 2 and gen["depth"] == 2
        assert gen["nbr_rank"] == ch["nbr_rank"] and gen["nbr_layer"] == ch["nbr_layer"]
        assert gen["recv_counts"] == ch["recv_counts"] and gen["nghost1"] == n1
        assert ch["nbr_layer"] == sorted(ch["nbr_layer"])                   # layer-1 entries first
        roff = np.concatenate([[0], np.cumsum(ch["recv_counts"])])
        for i in range(len(ch["nbr_rank"])):
            assert set(gen["gid"][nie + roff[i]:nie + roff[i + 1]]) == set(ch["gid"][nie + roff[i]:nie + roff[i + 1]])
            assert set(gen["gid"][gen["send_lists"][i]]) == set(ch["gid"][ch["send_lists"][i]])
            # a pair's tets are ordered by global id on both sides
            seg = ch["gid"][nie + roff[i]:nie + roff[i + 1]]
            assert (np.diff(seg) > 0).all() and (np.diff(ch["gid"][ch["send_lists"][i]]) > 0).all()
        assert np.array_equal(ch["gid"][:nie + n1], one[r]["gid"])          # the depth-1 chunk is the prefix
        # layers against a breadth-first search over the undivided mesh's face adjacency
        mine = part_g == r
        d1 = np.zeros(len(part_g), dtype=bool)
        for e in np.nonzero(mine)[0]:
            for nb in esuel[e]:
                if nb >= 0 and not mine[nb]:
                    d1[nb] = True
        d2 = np.zeros(len(part_g), dtype=bool)
        for e in np.nonzero(d1)[0]:
            for nb in esuel[e]:
                if nb >= 0 and not mine[nb] and not d1[nb]:
                    d2[nb] = True
        assert set(g["gid"][d1]) == set(ch["gid"][nie:nie + n1])
        assert set(g["gid"][d2]) == set(ch["gid"][nie + n1:])
    for fam in (chunks, gens):
        for r, ch in enumerate(fam):
            nie = ch["nielem"]
            roff = np.concatenate([[0], np.cumsum(ch["recv_counts"])])
            for i, (q, l) in enumerate(zip(ch["nbr_rank"], ch["nbr_layer"])):
                o = fam[q]
                j = [k for k, (qq, ll) in enumerate(zip(o["nbr_rank"], o["nbr_layer"])) if qq == r and ll == l]
                assert len(j) == 1
                assert np.array_equal(ch["gid"][nie + roff[i]:nie + roff[i + 1]], o["gid"][o["send_lists"][j[0]]])
    if parts == (2, 2, 2):          # the edge-diagonal ranks appear in layer 2 only
        assert any(l == 2 and q not in [qq for qq, ll in zip(c["nbr_rank"], c["nbr_layer"]) if ll == 1]
                   for c in chunks for q, l in zip(c["nbr_rank"], c["nbr_layer"]))


def test_depth_one_chunk_is_unchanged_by_the_depth_argument():
    from quinoa_amd import meshgen, partition
    g = meshgen.kuhn_box(5, 4, 3)
    part = partition.partition(g["coord"], g["inpoel"], 3, "rcb")
    a = partition.build_chunk(g["coord"], g["inpoel"], g["sidesets"], part, 3, 1)
    b = partition.build_chunk(g["coord"], g["inpoel"], g["sidesets"], part, 3, 1, depth=1)
    assert a["nbr_layer"] == [1] * len(a["nbr_rank"]) and a["nghost1"] == len(a["gid"]) - a["nielem"]
    for k in ("gid", "inpoel", "coord"):
        assert np.array_equal(a[k], b[k])
    with pytest.raises(Exception):
        partition.build_chunk(g["coord"], g["inpoel"], g["sidesets"], part, 3, 1, depth=3)
