#!/usr/bin/env python3
"""bench.py -- DG-P1 CompFlow element-updates/s on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--nx NX] [--no-north-star]

Workload (config.workload): BASELINE.json configs[1] -- CompFlow Euler Sod
shock tube, DG-P1 (dgp1), HLLC, Superbee limiter, CFL 0.3, on a synthetic
unstructured Kuhn-tet box of NX^3 hexes PER GPU (default 55 -> 998,250 tets),
extrapolate on the x faces, symmetry elsewhere.  A "step" is one SSP-RK3 time
step of the reference's loop: 3 x (halo, limiter, halo, [stage 0: CFL dt +
min-reduce], RHS, RK update), all on device-resident fields.  One
element-update = one tet through one RK stage's RHS evaluation, so
value = tets * 3 * K / t.  N > 1: one process per GPU (torch.distributed,
backend nccl = RCCL), block decomposition with one-layer ghost halo exchanged
point-to-point, weak scaling (fixed tets per GPU).

The same run then times the NORTH-STAR POINT (BASELINE.json north_star: DG-P1 RHS at
~10 M tets; >= 6x strong scaling 1 -> 8): the same physics on a fixed-size 119^3 x 6 =
10 110 954-tet box cut across the N ranks -- `north_star_point` in the JSON line, i.e.
the roofline fraction of the RHS kernel at 10 M tets at N = 1 and a STRONG-scaling
value for every N (`--no-north-star` skips it, `--strong-nx M` changes the box).
BASELINE.md section 5: 5 warm-up + 50 timed steps (the defaults).
At N = 1 it also times config 5's refine / re-upload loop (`amr_point`: a small box incl. a
decomposition, and the north-star box refined 10.1 M -> 80.9 M tets) and config 3 at
its own size, vortical flow DG-P2 + WENO on 7 986 000 tets (`config3_point`), and one GPU's
7 986 000-tet share of config 4's Sedov run (`config4_point`).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# this pool's host driver only supports dmabuf IPC (RCCL / device-tensor sharing)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# Second reading beside the HBM roofline (the as-written algorithm sits at the fp64 ridge): the
# kernels' vector-instruction counts against the chip's issue rate.  A wave64 fp64 instruction occupies its SIMD
# for 4 cycles (tools/ubench_fp64.hip: 4.2 measured), so 1024 SIMDs at the 2.4 GHz peak clock issue at most
# 614.4 G wave instructions per second.  Instructions per tet from the committed PMC passes (SQ_INSTS_VALU per
# launch / tets; which files: profiles/pmc_index.json).
VALU_ISSUE_PEAK_G = 1024 * 2.4 / 4.0


def valu_instr_per_tet(kind):
    """Vector wave-instructions per tet of the RHS kernels of order `kind` ("p1" / "p2"), read at run time from
    the committed counter file profiles/pmc_index.json points at (SQ_INSTS_VALU per launch, launch-weighted
    over the kernel's instantiations, / the tets of the profiled mesh) -- so the figure follows the kernels
    that were last profiled, not a constant in this file.  None when the files are missing."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_index.json")) as fh:
            ent = json.load(fh)[kind]
        with open(os.path.join(ROOT, "profiles", ent["file"])) as fh:
            pmc = json.load(fh)
        tot = w = 0.0
        for k, wt in ent["kernels"].items():
            tot += wt * pmc[k]["SQ_INSTS_VALU"]; w += wt
        return tot / w / ent["tets"], "profiles/" + ent["file"]
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None, None


def valu_reading(kind, tets, avg_ms):
    ipt, src = valu_instr_per_tet(kind)
    if ipt is None:
        return None
    ach = ipt * tets / (avg_ms * 1e-3) / 1e9
    return {"wave_instructions_per_tet": ipt, "achieved_G_wave_instr_per_s": ach,
            "issue_peak_G_wave_instr_per_s": VALU_ISSUE_PEAK_G, "frac_of_issue_peak": ach / VALU_ISSUE_PEAK_G,
            "source": "SQ_INSTS_VALU of the committed rocprofv3 --pmc pass %s (read at run time through "
                      "profiles/pmc_index.json); a wave64 fp64 instruction holds its SIMD for 4 cycles" % src}


def cpu_baseline(budget_s=10.0, budget_all_s=5.0):
    """The CPU oracle (plain-C restatement of the reference's loops) timed on a bounded
    sample of the same workload: on 1 core (the headline fields), and on all host cores
    with one independent mesh partition per thread and no halo -- the closest analogue of
    the reference's one-chare-per-PE run; ctypes releases the GIL in the C call."""
    import threading
    import numpy as np
    from oracle import oracle as O
    from quinoa_amd import meshgen
    n = 24
    ch = meshgen.kuhn_box(n, n, n)

    def make():
        om = O.OracleMesh(ch["coord"], ch["inpoel"], ch["sidesets"])
        cfg = O.make_cfg(4, flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4)
        orc = O.Oracle(om, cfg, bc_sym=[3, 4, 5, 6], bc_extrapolate=[1, 2])
        Lm = orc.lhs()
        U = orc.initialize(Lm, 0.0)
        work = (np.zeros_like(U), np.zeros(om.nelem * orc.npropr))
        return om, orc, Lm, U, work

    om, orc, Lm, U, work = make()
    orc.step(0.0, U, Lm, cfl=0.3, work=work)          # warm-up
    t0, steps = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s:
        orc.step(0.0, U, Lm, cfl=0.3, work=work)
        steps += 1
    el = time.perf_counter() - t0
    out = {"value": om.nelem * 3 * steps / el / 1e6, "unit": "M element-updates/s",
           "cores": 1, "kind": "port",
           # the reference's own DG-P1 RHS kernels timed in isolation on one
           # Xeon core (the reference itself cannot be built in this image; quoted, not measured here)
           "reference_kernels_M_per_s_per_core": 0.37,
           "sample": "oracle/dg_oracle.c, Sod DG-P1+Superbee CFL 0.3, %d-tet Kuhn box, %d full "
                     "RK3 steps in %.1f s" % (om.nelem, steps, el)}
    # all cores: one partition per thread
    ncore = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ncore = max(1, min(ncore, 16))        # the GPU box grants a 16-CPU share per GPU
    if ncore > 1:
        parts = [make() for _ in range(ncore)]
        per = max(1, int(steps * budget_all_s / el))
        def run(p):
            _, o, L_, U_, w_ = p
            for _ in range(per):
                o.step(0.0, U_, L_, cfl=0.3, work=w_)
        th = [threading.Thread(target=run, args=(p,)) for p in parts]
        t1 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        ela = time.perf_counter() - t1
        out["all_cores"] = {"value": om.nelem * 3 * per * ncore / ela / 1e6, "cores": ncore,
                            "sample": "%d threads, one %d-tet partition each (no halo), %d steps in %.1f s"
                                      % (ncore, om.nelem, per, ela)}
    return out


def run_workload(args, rank, world, local_rank, dims, lengths, parts, steps, warmup, comm_kind, use_dist, develop):
    """One run of the DG-P1 workload on the box `dims` (global hexes per axis) cut into `parts`:
    `warmup` untimed steps, `steps` timed steps from the initial state (the COLD-START reading),
    `develop` untimed steps that let the flow develop, then the `steps` timed steps of the headline,
    bracketed by barrier + synchronize, with nothing but the step in the loop; last, a short loop with
    HIP events around every RHS launch, halo exchange and all-reduce (qdg_profile_*).  Returns the
    measurements of this rank's chunk (max / sum over ranks done)."""
    import numpy as np
    import torch
    from quinoa_amd import capi, dg, meshgen
    self_halo = args.self_halo
    # two ghost layers by default (a rank limits its layer-1 ghosts itself: 3 halo exchanges per step, not 6)
    depth = args.halo_depth if (world > 1 or self_halo) else 1
    ch = meshgen.kuhn_box_chunk(dims[0], dims[1], dims[2], lengths=lengths, parts=parts, rank=rank, depth=depth)
    if self_halo:
        ch["nbr_rank"] = [0 for _ in ch["nbr_rank"]]
        # (the rank is its own neighbour in name only: every entry sends as many rows as it receives)
        ch["send_lists"] = [np.resize(s_, n_) for s_, n_ in zip(ch["send_lists"], ch["recv_counts"])]
    nielem = int(ch["nielem"])
    opts = {"graph_step": 1 if args.graph else 0, "halo_depth": depth}
    if args.workload == "sedov":     # config 4's physics (symmetry on x-min, y-min and the z faces)
        ctx = capi.Context(4, flux="hllc", limiter="superbeep1", problem="sedov_blastwave", gamma=1.4,
                           cfl=0.3, bc_extrapolate=[2, 4], bc_sym=[1, 3, 5, 6], device=local_rank, options=opts)
    else:
        ctx = capi.Context(4, flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4,
                           cfl=0.3, bc_extrapolate=[1, 2], bc_sym=[3, 4, 5, 6], device=local_rank, options=opts)
    # FaceData, geometry and the device layout of the chunk (with its ghost layer) are built on
    # the GPU: only connectivity, coordinates and side-set triangles cross PCIe
    # ... with the tets' global ids: faces oriented by global id, so the N-rank run takes the branches of the
    # 1-rank run wherever HLLC falls through to the stored right state (qdg_mesh_from_chunk_gid)
    mesh = capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"], nielem=nielem,
                                       elem_gid=None if self_halo else ch["gid"])
    comm = None
    if use_dist:
        if comm_kind == "rccl":
            # the product transport; a rank that cannot load RCCL ends the run (no silent
            # degradation to another transport: ask for it with --comm torch)
            comm = dg.RcclComm(ctx)
        else:
            comm = dg.TorchComm()
    drv = dg.DGDriver(ctx, mesh, ch["nbr_rank"], ch["send_lists"], ch["recv_counts"], comm,
                      nghost1=ch["nghost1"] if depth == 2 else 0)
    mesh.state_initialize(0.0)

    # outside the timed region: total mass and energy before / after the run.  Between
    # symmetry walls and (still undisturbed) extrapolation faces they are conserved, and a
    # flux that does not match across a chunk boundary would show up here.
    vol = _tet_volumes(ch["coord"], ch["inpoel"][:nielem])

    def totals():
        Uh = mesh.state_download().reshape(-1, 20)[:nielem]
        v = np.array([(Uh[:, 0] * vol).sum(), (Uh[:, 16] * vol).sum()])
        if world > 1:
            tv = torch.tensor(v, dtype=torch.float64, device="cuda")
            torch.distributed.all_reduce(tv, op=torch.distributed.ReduceOp.SUM)
            v = tv.cpu().numpy()
        return v

    tot0 = totals()

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            torch.distributed.barrier()

    def timed(n):
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            drv.step(0.0)
        sync()
        return time.perf_counter() - t0

    for _ in range(warmup):
        drv.step(0.0)
    el_cold = timed(steps) if develop > 0 else None        # the first steps after the initial discontinuity
    t_flow = 0.0
    for i in range(develop):                               # untimed: the flow develops
        drv.step(0.0)
        if (i & 255) == 255:
            t_flow += 256 * drv.dt_taken()                  # (dt varies slowly: sampled, for the record only)
    el = timed(steps)                                       # THE measurement: nothing but the step in the loop
    # the flow-independent bound: the same steps with the limiter writing back EVERY tile (it normally skips the
    # tiles it leaves unchanged) -- what a flow that is developed everywhere would run at; identical results
    ctx.set_option("limiter_write_all", 1); ctx.set_option("graph_step", 0)
    el_all = timed(steps)
    ctx.set_option("limiter_write_all", 0); ctx.set_option("graph_step", opts["graph_step"])
    # a second, short loop with events on the library's stream around every RHS launch, exchange and all-reduce
    nprof = max(5, min(steps, 10))
    mesh.profile_enable(True)
    for _ in range(nprof):
        drv.step(0.0)
    sync()
    pr = mesh.profile_read_all() if hasattr(mesh, "profile_read_all") else None
    mesh.profile_enable(False)
    nl, ms = pr["rhs"]
    dt_last = drv.dt_taken()
    if not (dt_last > 0.0 and np.isfinite(dt_last)):
        raise SystemExit("invalid run: dt = %r" % dt_last)
    drift = np.abs(totals() - tot0) / np.abs(tot0)
    if not self_halo and not (drift.max() <= 1e-9):
        raise SystemExit("invalid run: mass / energy drift %r (halo or flux mismatch)" % (drift,))
    ntet = nielem
    # what the transport itself reports: ranks in the RCCL communicator (ncclCommCount), this rank's device
    seen = None
    graph = None
    if isinstance(comm, dg.RcclComm):
        seen = comm.comm.info()
        st, ng, nrep, err = mesh.step_graph_status()
        graph = {"state": {0: "not tried", 1: "in use", -1: "refused"}.get(st, st), "graphs": ng, "replays": nrep,
                 "error": err or None}
    per_rank = [{"rank": rank, "tets": nielem, "ghost_tets": int(len(ch["gid"]) - nielem),
                 "neighbours": [int(r) for r in ch["nbr_rank"]], "device": local_rank,
                 "rhs_avg_launch_ms": ms / max(nl, 1),
                 # the communication of a step where it happens (events around every exchange = pack + grouped
                 # ncclSend / ncclRecv incl. the gaps in front of them, and around the dt all-reduce)
                 "exchanges_per_step": pr["halo"][0] / nprof, "halo_ms_per_step": pr["halo"][1] / nprof,
                 "allreduces_per_step": pr["allreduce"][0] / nprof,
                 "allreduce_ms_per_step": pr["allreduce"][1] / nprof,
                 "step_graph": graph,
                 "halo_plan": dict(zip(("entries", "layer1_ghosts_limited_here", "packs_folded_into_producers"),
                                       mesh.halo_info()))}]
    if world > 1:
        tt = torch.tensor([el, float(ntet), el_cold or 0.0, el_all], dtype=torch.float64, device="cuda")
        mx = tt.clone(); torch.distributed.all_reduce(mx, op=torch.distributed.ReduceOp.MAX)
        sm = tt.clone(); torch.distributed.all_reduce(sm, op=torch.distributed.ReduceOp.SUM)
        el, ntet = float(mx[0]), int(round(float(sm[1])))
        el_cold = float(mx[2]) if el_cold is not None else None
        el_all = float(mx[3])
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
    res = {"el": el, "el_cold": el_cold, "el_all": el_all, "ntet": ntet, "ntet_local": nielem, "avg_ms": ms / max(nl, 1), "launches": nl,
           "alg": mesh.rhs_algorithmic_bytes(), "dt_last": dt_last, "drift": drift, "develop": develop, "t_flow": t_flow,
           "backend": None if comm is None else comm.backend, "per_rank": per_rank, "halo_depth": depth,
           "ranks_seen_by_rccl": None if seen is None else seen[0]}
    mesh.close()
    if isinstance(comm, dg.RcclComm):
        comm.close()
    ctx.close()
    return res


def _tet_volumes(coord, inpoel):
    """tet volumes (tk::triple / 6, src/Mesh/DerivedData.cpp:genGeoElemTet) for the conservation check"""
    import numpy as np
    c = np.asarray(coord, dtype=np.float64)
    t = np.asarray(inpoel).reshape(-1, 4)
    a, b, d = c[t[:, 1]] - c[t[:, 0]], c[t[:, 2]] - c[t[:, 0]], c[t[:, 3]] - c[t[:, 0]]
    return np.einsum("ij,ij->i", a, np.cross(b, d)) / 6.0


def amr_point(local_rank, nx=32, steps=20, with_partition=True, resident=True, reserve=False):
    """BASELINE config 5's loop once, on one GPU: Sod DG-P1 on an nx^3 Kuhn box, `steps` time
    steps, uniform 1:8 refinement (the refinement the reference's DG scheme performs during
    time stepping), mesh-derived data of the new mesh on the device, state handed over on the
    device, `steps` more steps.  Reports what the re-partition / re-upload step costs."""
    import numpy as np
    from quinoa_amd import amr, capi, meshgen
    ch = meshgen.kuhn_box(nx, nx, nx)
    ctx = capi.Context(4, flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4,
                       cfl=0.3, bc_extrapolate=[1, 2], bc_sym=[3, 4, 5, 6], device=local_rank)
    run = amr.RefinedRun(ctx, ch["coord"], ch["inpoel"], ch["sidesets"], resident=resident)
    run.mesh.state_initialize(0.0)
    ne0 = run.mesh.nielem
    reserved = 0
    if reserve:
        # a run that knows it will re-mesh takes its memory budget from the driver ONCE, at start-up
        # (qdg_device_pool_reserve; here 60 % of what is free): the re-mesh then allocates from that region
        t0 = time.perf_counter()
        reserved = int(0.6 * ctx.device_memory()[0])
        ctx.reserve_device_memory(reserved)
        t_reserve = time.perf_counter() - t0

    def advance(n):
        run.mesh.step(0.0, want_dt=False)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            run.mesh.step(0.0, want_dt=False)
        ctx.synchronize()
        return (time.perf_counter() - t0) / n

    ms0 = advance(steps) * 1e3
    th, tr, tt = run.refine()
    ms1 = advance(steps) * 1e3
    U = run.mesh.state_download()
    ne1 = run.mesh.nielem
    ok = bool(np.isfinite(U).all())
    deref = None
    if resident:
        # ... and back: config 5's "deref" -- the 8:1 coarsening of the refined chunk on the device
        # (qdg_mesh_derefine_uniform, conservative means), then `steps` steps on the coarse mesh again
        del U
        ctx.synchronize()
        t0 = time.perf_counter()
        mc = run.mesh.derefine_uniform("mean")
        ctx.synchronize()
        t_deref = time.perf_counter() - t0
        run.mesh.close()
        run.mesh = mc
        ms2 = advance(steps) * 1e3
        deref = {"derefine_total_ms": t_deref * 1e3, "tets_after": int(mc.nielem), "ms_per_step_after": ms2,
                 "finite": bool(np.isfinite(mc.state_download()).all()),
                 "note": "qdg_mesh_derefine_uniform: coarse connectivity from the children, boundary faces from the refined "
                         "ones, general device build (face sort), parent state = volume-weighted mean of its children"}
    run.mesh.close(); ctx.close()
    out = _amr_single(nx, steps, ne0, ne1, ms0, ms1, th, tr, tt, ok)
    if deref is not None:
        out["derefine"] = deref
    if resident:
        out["remesh_total_ms"] = (th + tr + tt) * 1e3
        out["steps_of_new_mesh_per_remesh"] = (th + tr + tt) * 1e3 / ms1
        out["host_copy_complete_ms"] = run.host_copy_s * 1e3
        out["state_buffers_reserved_ahead"] = bool(reserve)
        if reserve:
            out["reserve_bytes"], out["reserve_ms_outside_the_remesh"] = reserved, t_reserve * 1e3
        out["note"] = ("qdg_mesh_refine_uniform: the whole re-mesh in ONE call on the device (key rebuild_upload_ms = "
                       "remesh_total_ms): refinement from the connectivity the handle keeps resident (edge sort, midpoints, "
                       "children), esuel of the children by the 1:8 template, boundary faces from the parents', faces + "
                       "geometry, Morton order, numbering, face tasks, state buffers, state child <- parent; nothing is "
                       "uploaded.  host_copy_complete_ms = from the start of that call until the host holds the refined "
                       "mesh for its book-keeping (second thread, second stream, off the critical path)")
    if with_partition:
        out = {"on_a_decomposition": amr_partitioned(local_rank, ch, nparts=2, steps=5), **out}
    return out


def amr_partitioned(local_rank, g, nparts, steps, host_path=True, reserve=False):
    """The same re-mesh on a decomposition: `nparts` chunks with ghost halos (all on this one GPU,
    quinoa_amd.dg.LocalChunks), every chunk refined by its own rank's logic, without communication.
    Device path (round 4): qdg_mesh_refine_chunk -- children, new ghost layer, halo plan, chunk build with its
    ghosts, halo set-up and the owned state in ONE call from the connectivity / global ids / plan the handle
    keeps.  Host path (rounds 2-3, timed beside it when host_path): qdg_refine_chunk on the host's cores,
    qdg_mesh_from_chunk on the device, qdg_state_transfer.  Times are the maximum over the chunks, i.e. what a
    rank of a one-process-per-GPU run spends."""
    import numpy as np
    from quinoa_amd import amr, capi, dg, partition
    ctx = capi.Context(4, flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4,
                       cfl=0.3, bc_extrapolate=[1, 2], bc_sym=[3, 4, 5, 6], device=local_rank,
                       options={"keep_connectivity": 1})
    if reserve:       # the run's memory budget from the driver, once (see amr_point)
        ctx.reserve_device_memory(int(0.6 * ctx.device_memory()[0]))
    pt = partition.partition(g["coord"], g["inpoel"], nparts, "rcb")
    chunks = [partition.build_chunk(g["coord"], g["inpoel"], g["sidesets"], pt, nparts, r) for r in range(nparts)]

    def build(ch):
        return capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"], nielem=ch["nielem"],
                                           elem_gid=ch["gid"])

    meshes = [build(ch) for ch in chunks]
    for m in meshes:
        m.state_initialize(0.0)
    drv = dg.LocalChunks(ctx, meshes, chunks)
    for _ in range(steps):
        drv.step(0.0)
    ctx.synchronize()
    t_ref = t_reb = t_tr = 0.0
    if host_path:
        for ch, m in zip(chunks, meshes):
            t0 = time.perf_counter()
            ch2, par = amr.refine_chunk(ch)
            t1 = time.perf_counter()
            m2 = build(ch2)
            ctx.synchronize()
            t2 = time.perf_counter()
            amr.state_transfer(m, m2, par)
            ctx.synchronize()
            t3 = time.perf_counter()
            m2.close()
            t_ref, t_reb, t_tr = max(t_ref, t1 - t0), max(t_reb, t2 - t1), max(t_tr, t3 - t2)
    t_dev = 0.0
    new_chunks, new_meshes = [], []
    for ch, m in zip(chunks, meshes):
        t0 = time.perf_counter()
        m2, plan = m.refine_chunk(ch["nbr_rank"])
        ctx.synchronize()
        t_dev = max(t_dev, time.perf_counter() - t0)
        m.close()
        new_chunks.append(plan); new_meshes.append(m2)
    drv = dg.LocalChunks(ctx, new_meshes, new_chunks)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        drv.step(0.0)
    ctx.synchronize()
    ms_step = (time.perf_counter() - t0) / steps * 1e3
    ok = all(bool(np.isfinite(m.state_download()).all()) for m in new_meshes)
    out = {"chunks": nparts, "owned_tets_per_chunk_before": [int(c["nielem"]) for c in chunks],
           "owned_tets_per_chunk_after": [int(c["nielem"]) for c in new_chunks],
           "ghost_tets_per_chunk_after": [int(len(c["gid"]) - c["nielem"]) for c in new_chunks],
           "remesh_device_ms": t_dev * 1e3, "ms_per_step_after_all_chunks_on_this_gpu": ms_step, "finite": ok,
           "memory_reserved_ahead": bool(reserve),
           "note": "remesh_device_ms = qdg_mesh_refine_chunk per rank (maximum over the chunks): refinement, new "
                   "ghost layer and halo plan, chunk build with ghosts, halo set-up, owned state -- one call, nothing "
                   "uploaded; the host gets global ids, parents and the plan back"}
    if host_path:
        out.update({"refine_chunk_host_ms": t_ref * 1e3, "rebuild_with_ghosts_device_ms": t_reb * 1e3,
                    "state_transfer_ms": t_tr * 1e3})
    for m in new_meshes:
        m.close()
    ctx.close()
    return out


def amr_one_rank(local_rank, nx, parts=(2, 1, 1), rank=0, depth=2):
    """What ONE rank of a decomposition spends on a re-mesh, measured with only that rank's chunk on the GPU (as in a
    one-process-per-GPU run: the other chunks' buffers do not compete for this device's memory): the rank's chunk of
    the nx^3 box cut into `parts`, with `depth` ghost layers, built on the device with its halo plan, state
    initialised; then qdg_mesh_refine_chunk -- refinement, new ghost layer(s) and halo plan, chunk build with ghosts,
    halo set-up, owned state: one call, nothing uploaded."""
    import numpy as np
    from quinoa_amd import capi, meshgen
    ch = meshgen.kuhn_box_chunk(nx, nx, nx, parts=parts, rank=rank, depth=depth)
    ctx = capi.Context(4, flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4, cfl=0.3,
                       bc_extrapolate=[1, 2], bc_sym=[3, 4, 5, 6], device=local_rank,
                       options={"keep_connectivity": 1, "halo_depth": depth})
    ctx.reserve_device_memory(int(0.6 * ctx.device_memory()[0]))
    mesh = capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"], nielem=ch["nielem"],
                                       elem_gid=ch["gid"])
    mesh.halo_setup(ch["nbr_rank"], ch["send_lists"], ch["recv_counts"])
    if depth == 2:
        mesh.halo_set_depth(ch["nghost1"])
    mesh.state_initialize(0.0)
    ctx.synchronize()
    t0 = time.perf_counter()
    m2, plan = mesh.refine_chunk()
    ctx.synchronize()
    t_dev = time.perf_counter() - t0
    mesh.close()
    ok = bool(np.isfinite(m2.state_download()[:20 * plan["nielem"]]).all())
    out = {"cut": "%dx%dx%d, rank %d, %d ghost layer(s)" % (parts + (rank, depth)),
           "owned_tets_before": int(ch["nielem"]), "ghost_tets_before": int(len(ch["gid"]) - ch["nielem"]),
           "owned_tets_after": int(plan["nielem"]), "ghost_tets_after": int(len(plan["gid"]) - plan["nielem"]),
           "plan_entries_after": list(zip(plan["nbr_rank"], plan["nbr_layer"])),
           "remesh_device_ms": t_dev * 1e3, "finite": ok, "memory_reserved_ahead": True,
           "note": "qdg_mesh_refine_chunk of ONE rank's chunk, alone on the GPU, from a reserved memory region"}
    m2.close(); ctx.close()
    return out


def _amr_single(nx, steps, ne0, ne1, ms0, ms1, th, tr, tt, ok):
    return {"workload": "Sod DG-P1 + Superbee, %d^3 box: %d steps, uniform 1:8 refinement, %d steps" % (nx, steps, steps),
            "tets_before": ne0, "tets_after": ne1, "ms_per_step_before": ms0, "ms_per_step_after": ms1,
            "refine_host_ms": th * 1e3, "rebuild_upload_ms": tr * 1e3, "state_transfer_ms": tt * 1e3,
            "rebuild_upload_ms_per_Mtet": tr * 1e3 / (ne1 / 1e6),
            "steps_of_new_mesh_per_rebuild": tr * 1e3 / ms1, "finite": ok,
            "note": "refine (key refine_host_ms) = qdg_refine_uniform_device: edge sort + scans on the GPU, the "
                    "refined mesh copied back to the host, where AMR bookkeeping stays; rebuild = "
                    "qdg_mesh_from_connectivity on the refined mesh, all on the device: boundary faces, "
                    "FaceData, geometry, Morton order, numbering, face tasks (only connectivity + "
                    "coordinates cross PCIe); transfer = qdg_state_transfer (child <- parent, device). "
                    "The rebuild includes hipMalloc of the new mesh's buffers: ~34 ms per GiB on VRAM that any "
                    "process on the box has used before (driver scrubbing; DESIGN.md section 6)"}


def config3_point(local_rank, nx=110, steps=20):
    """BASELINE config 3 at its own size on one GPU: vortical_flow DG-P2 + wenop1 (alpha 0.1,
    beta 1, p0 10, gamma 5/3, Dirichlet on all six sides), nx^3 x 6 tets, prescribed dt
    (1e-5 scaled with the mesh size), `steps` timed SSP-RK3 steps after two warm-up steps."""
    import numpy as np
    from quinoa_amd import capi, meshgen
    ch = meshgen.kuhn_box(nx, nx, nx)
    ctx = capi.Context(10, flux="hllc", limiter="wenop1", problem="vortical_flow", gamma=5.0 / 3.0,
                       cweight=1.0, alpha=0.1, beta=1.0, p0=10.0, dt=1e-5 * 10.0 / nx,
                       bc_dirichlet=[1, 2, 3, 4, 5, 6], device=local_rank)
    mesh = capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"])
    mesh.state_initialize(0.0)
    el, nl, ms = _time_single_chunk(ctx, mesh, steps, warm=2)
    alg = mesh.rhs_algorithmic_bytes()
    U = mesh.state_download()
    ok = bool(np.isfinite(U).all())
    ne = mesh.nielem
    mesh.close(); ctx.close()
    ach = alg / (ms / nl * 1e-3) / 1e9
    vr = valu_reading("p2", ne, ms / nl)
    # the roof that BINDS this kernel is the fp64 vector issue rate, not HBM (VALU ~80 % busy at
    # 2 waves per SIMD, 2.3x the algorithmic bytes fetched at < 3 TB/s): report it as a roofline of its own
    binding = None if vr is None else {
        "bound": "fp64_valu", "achieved": vr["achieved_G_wave_instr_per_s"], "peak": VALU_ISSUE_PEAK_G,
        "unit": "G wave64 fp64-instructions/s", "frac": vr["frac_of_issue_peak"],
        "instructions_per_tet": vr["wave_instructions_per_tet"], "source": vr["source"]}
    return {"workload": "CompFlow vortical_flow DG-P2 + wenop1, Kuhn-tet box %d^3 hexes = %d tets, prescribed dt, "
                        "%d timed steps" % (nx, ne, steps),
            "tets_total": ne, "steps": steps, "value": ne * 3 / el / 1e6, "unit": "M element-updates/s",
            "ms_per_step": el * 1e3, "finite": ok,
            "roofline": {"bound": "hbm", "kernel": "qdg::k_rhs_p2s (lane pair per tet; RK update fused in; 837 B/tet "
                                                   "counted per launch)",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "avg_launch_ms": ms / nl, "launches": nl, "algorithmic_bytes_per_launch": alg,
                         "traffic": None, "fp64_issue": vr},
            "roofline_binding": binding}


def _time_single_chunk(ctx, mesh, steps, warm=2, develop=0):
    """`warm` + `develop` untimed steps, `steps` timed steps with nothing but the step in the loop, then a short
    loop with HIP events around the RHS launches; -> (seconds per step, profiled launches, their summed ms)
    [+ seconds per step of the first `steps` steps when develop > 0]"""
    def timed(n):
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            mesh.step(0.0, want_dt=False)
        ctx.synchronize()
        return (time.perf_counter() - t0) / n
    for _ in range(warm):
        mesh.step(0.0, want_dt=False)
    cold = timed(steps) if develop > 0 else None
    for _ in range(develop):
        mesh.step(0.0, want_dt=False)
    el = timed(steps)
    mesh.profile_enable(True)
    for _ in range(max(3, min(steps, 8))):
        mesh.step(0.0, want_dt=False)
    ctx.synchronize()
    nl, ms = mesh.profile_read()
    mesh.profile_enable(False)
    return (el, nl, ms) if cold is None else (el, nl, ms, cold)


def config4_point(local_rank, nx=110, steps=20, develop=2000):
    """One GPU's share of BASELINE config 4 (Sedov blast DG-P1 + Superbee, CFL 0.3, 64 M tets over
    8 GPUs): nx^3 x 6 = 7 986 000 tets on this GPU, no halo (the 8-GPU run is `--gpus 8 --nx 110`);
    timed after `develop` untimed steps (the blast has left its corner), cold-start rate beside it."""
    import numpy as np
    from quinoa_amd import capi, meshgen
    ch = meshgen.kuhn_box(nx, nx, nx)
    ctx = capi.Context(4, flux="hllc", limiter="superbeep1", problem="sedov_blastwave", gamma=1.4, cfl=0.3,
                       bc_sym=[1, 3, 5, 6], bc_extrapolate=[2, 4], device=local_rank)
    mesh = capi.mesh_from_connectivity(ctx, ch["inpoel"], ch["coord"], ch["sidesets"])
    mesh.state_initialize(0.0)
    r = _time_single_chunk(ctx, mesh, steps, warm=2, develop=develop)
    el, nl, ms = r[:3]
    cold = r[3] if len(r) > 3 else None
    alg = mesh.rhs_algorithmic_bytes()
    U = mesh.state_download()
    ok = bool(np.isfinite(U).all())
    ne = mesh.nielem
    mesh.close(); ctx.close()
    ach = alg / (ms / nl * 1e-3) / 1e9
    return {"workload": "CompFlow Sedov blast wave DG-P1 + superbeep1, CFL 0.3, Kuhn-tet box %d^3 hexes = %d tets "
                        "(one GPU's share of config 4), %d timed steps after %d untimed ones" % (nx, ne, steps, develop),
            "tets_total": ne, "steps": steps, "value": ne * 3 / el / 1e6, "unit": "M element-updates/s",
            "ms_per_step": el * 1e3, "finite": ok,
            "cold_start_M_per_s": None if cold is None else ne * 3 / cold / 1e6,
            "roofline": {"bound": "hbm", "kernel": "qdg::k_rhs_p1w", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "avg_launch_ms": ms / nl, "launches": nl,
                         "algorithmic_bytes_per_launch": alg, "traffic": None}}


def real_mesh_point(local_rank, levels=(2, 3), steps=20, develop=500):
    """A genuinely unstructured mesh: the reference's own 31 304-tet unit-cube fixture (unitcube_01_31k.exo of
    its SlotCyl / Sod regression cases, kept as data in tests/golden/slot_cyl_dg.npz), refined 1:8 on the device
    `levels` times (2.0 M / 16.0 M tets) -- valence, nodes per tet and tile surface are a mesh generator's, not
    the Kuhn box's.  Sod DG-P1 + Superbee, CFL 0.3: step rate, RHS roofline fraction, face-task mix."""
    import numpy as np
    from quinoa_amd import capi
    fx = np.load(os.path.join(ROOT, "tests", "golden", "slot_cyl_dg.npz"))
    coord, inpoel, tri = fx["coord"], fx["inpoel"], fx["ss_tri_1"]
    # the fixture lists the whole boundary as one side set: split it by cube face (1 x=0, 2 x=1, 3 y=0, ...)
    c = coord[tri].mean(axis=1)
    sets = {}
    for ax in range(3):
        sets[2 * ax + 1] = tri[c[:, ax] < 1e-9]
        sets[2 * ax + 2] = tri[c[:, ax] > 1.0 - 1e-9]
    out = []
    for lev in levels:
        ctx = capi.Context(4, flux="hllc", limiter="superbeep1", problem="sod_shocktube", gamma=1.4, cfl=0.3,
                           bc_extrapolate=[1, 2], bc_sym=[3, 4, 5, 6], device=local_rank,
                           options={"keep_connectivity": 1})
        mesh = capi.mesh_from_connectivity(ctx, inpoel, coord, sets)
        for _ in range(lev):
            new, _ref = mesh.refine_uniform(host_copy=False)
            mesh.close()
            mesh = new
        mesh.state_initialize(0.0)
        el, nl, ms, cold = _time_single_chunk(ctx, mesh, steps, warm=2, develop=develop)
        ne = mesh.nielem
        alg = mesh.rhs_algorithmic_bytes()
        ls = mesh.layout_stats()
        ok = bool(np.isfinite(mesh.state_download()).all())
        mesh.close(); ctx.close()
        ach = alg / (ms / nl * 1e-3) / 1e9
        out.append({"tets_total": ne, "refinements_1_to_8": lev, "steps": steps, "untimed_steps_first": develop,
                    "value": ne * 3 / el / 1e6, "unit": "M element-updates/s", "ms_per_step": el * 1e3,
                    "cold_start_M_per_s": ne * 3 / cold / 1e6, "finite": ok,
                    "face_tasks_per_tet": {"in_tile_evaluated_once": ls["in_tile"] / ne,
                                           "to_other_tiles": ls["to_other_tiles"] / ne, "boundary": ls["boundary"] / ne},
                    "roofline": {"bound": "hbm", "kernel": "qdg::k_rhs_p1w", "achieved": ach, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "avg_launch_ms": ms / nl, "launches": nl,
                                 "algorithmic_bytes_per_launch": alg, "traffic": None}})
    return {"workload": "CompFlow Sod DG-P1 + superbeep1, CFL 0.3, on the reference's unstructured unit-cube mesh "
                        "(31 304 tets, tests/golden/slot_cyl_dg.npz) refined 1:8 on the device", "points": out}


def self_launch(ngpus):
    """Run this script on `ngpus` ranks of this node through torch.distributed.run, as child processes
    sharing this process's stdout / stderr; returns the launcher's exit code."""
    import socket
    import subprocess
    try:                                     # (counting devices does not initialise the GPU on this image)
        import torch
        have = torch.cuda.device_count()
    except Exception:
        have = None
    if have is not None and have < ngpus:
        sys.stderr.write("bench.py --gpus %d: this node exposes %d GPU(s)\n" % (ngpus, have))
        return 2
    with socket.socket() as sk:               # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nx", type=int, default=55, help="hexes per direction PER GPU (weak scaling)")
    ap.add_argument("--develop", type=int, default=4000,
                    help="untimed steps between the cold-start timing and the headline timing, at --nx 55; other "
                         "box sizes take the steps that reach the same flow time (x nx/55).  The Superbee kernel "
                         "does not write back tiles it leaves unchanged, so the step is fastest right after the "
                         "initial discontinuity: the headline is taken on the developed flow, the cold-start "
                         "rate is reported beside it (0: headline = cold start, as before round 5)")
    ap.add_argument("--halo-depth", type=int, choices=[1, 2], default=2,
                    help="ghost layers of a rank's chunk (N > 1, --self-halo): 2 (default) = the rank limits its "
                         "layer-1 ghosts itself, 3 exchanges + 1 all-reduce per step; 1 = the reference's one layer, "
                         "6 exchanges + 1 all-reduce")
    ap.add_argument("--graph", action="store_true",
                    help="multi-rank / self-halo runs: qdg_step_comm replays its launch sequence (kernels + RCCL) as a "
                         "hipGraph (context option graph_step).  Measured on one GPU: <= 1 %% (the host is not the "
                         "bottleneck), so plain launches are the default")
    ap.add_argument("--no-real-mesh", action="store_true",
                    help="skip the point on the reference's unstructured 31 k-tet cube mesh refined to 2 M / 16 M tets")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["sod", "sedov"], default="sod",
                    help="physics of the DG-P1 run: the Sod shock tube of BASELINE config 2 (default, the "
                         "headline metric) or config 4's Sedov blast wave (`--gpus 8 --nx 110 --workload sedov` "
                         "is config 4: 63.9 M tets over 8 GPUs)")
    ap.add_argument("--no-north-star", action="store_true",
                    help="skip the fixed-size 10.1 M-tet north-star / strong-scaling point")
    ap.add_argument("--no-amr", action="store_true", help="skip the config-5 refinement point (N = 1 only)")
    ap.add_argument("--no-config3", action="store_true",
                    help="skip the config-3 point (DG-P2 + WENO at 7.99 M tets, N = 1 only)")
    ap.add_argument("--config3-nx", type=int, default=110)
    ap.add_argument("--no-config4", action="store_true",
                    help="skip the config-4 point (one GPU's 7.99 M-tet share of the Sedov run, N = 1 only)")
    ap.add_argument("--strong-nx", type=int, default=119,
                    help="hexes per direction of the fixed-size box of the strong-scaling point "
                         "(119 -> 10 110 954 tets, 220 -> 63 888 000 tets = config 4)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (nccl) even for one rank: exercises the "
                         "device-tensor slabs, the shared stream and the dt all-reduce")
    ap.add_argument("--comm", choices=["rccl", "torch"], default="rccl",
                    help="multi-rank transport: libqdg's own RCCL calls (default; the run FAILS if "
                         "RCCL cannot be loaded) or torch.distributed point-to-point")
    ap.add_argument("--self-halo", action="store_true",
                    help="one rank, chunk 0 of a 2x1x1 cut whose neighbour is the rank itself: "
                         "measures the halo machinery (pack, RCCL send/recv, unpack, all-reduce) "
                         "on a single GPU; not a physical set-up, prints the same JSON line")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by itself: start the N ranks as CHILD processes (one per GPU, the
        # launcher the driver would use) and relay rank 0's JSON line and the exit code -- the reference's
        # analogue is one command too (charmrun +pN inciter ..., cmake/test_runner.cmake:88-113).  Nothing in
        # this process has touched the GPU (torch is not even imported yet), and nothing is exec'ed.
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start one rank per GPU (python bench.py --gpus N does it "
                         "itself when WORLD_SIZE is unset)" % (args.gpus, world))
    import __graft_entry__
    __graft_entry__.ensure_built()          # no-op when libqdg.so is there
    import torch
    from quinoa_amd import meshgen

    use_dist = world > 1 or args.force_dist or args.self_halo
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    # ---- headline: weak scaling, NX^3 hexes per GPU ------------------------------------------
    parts = meshgen.parts_for(world) if not args.self_halo else (2, 1, 1)
    nx = args.nx
    def develop_for(n):                     # the same flow time on a finer / coarser box
        return int(round(args.develop * n / 55.0))
    w = run_workload(args, rank, world, local_rank, (nx * parts[0], nx * parts[1], nx * parts[2]),
                     (float(parts[0]), float(parts[1]), float(parts[2])), parts,
                     args.steps, args.warmup, args.comm, use_dist, develop_for(nx))
    # ---- north-star / strong-scaling point: fixed-size box cut across the ranks ---------------
    ns = None
    if not args.no_north_star and not args.self_halo:
        sx = args.strong_nx
        ns_steps, ns_warm = max(20, args.steps // 2), max(2, args.warmup // 2)   # >= 20 timed steps whatever --steps says
        ns = run_workload(args, rank, world, local_rank, (sx, sx, sx), (1.0, 1.0, 1.0), parts, ns_steps, ns_warm,
                          args.comm, use_dist, develop_for(sx))
        ns["steps"], ns["warmup"] = ns_steps, ns_warm

    if rank == 0:
        def roof(r):
            ach = r["alg"] / (r["avg_ms"] * 1e-3) / 1e9
            return ach, ach / HBM_PEAK_GBS
        achieved, frac = roof(w)
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get("nx") == nx and tj.get("n_gpus") == world:
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = ("profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                  "round %s on this workload; not re-measured in this run)" % tj.get("round"))
        out = {
            "metric": "M element-updates/sec, DG-P1 CompFlow on unstructured tets",
            "value": w["ntet"] * 3 * args.steps / w["el"] / 1e6,
            "unit": "M element-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": w["el"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("CompFlow Euler Sedov blast wave DG-P1 (dgp1, HLLC, superbeep1, "
                                    if args.workload == "sedov" else
                                    "CompFlow Euler Sod shock-tube DG-P1 (dgp1, HLLC, superbeep1, ")
                                   + "cfl 0.3), Kuhn-tet box %d^3 hexes per GPU" % nx
                                   + (", timed on the DEVELOPED flow: %d untimed steps first (flow time %.4f; the "
                                      "cold-start rate of the first steps is rates.cold_start_M_per_s)"
                                      % (w["develop"], w["t_flow"]) if w["develop"] else ", timed from the initial state"),
                       "tets_total": w["ntet"], "tets_per_gpu": w["ntet_local"],
                       "parallelism": "block decomposition %dx%dx%d, ghost-face halo" % parts
                                      + (", two ghost layers (3 exchanges per step)" if w["halo_depth"] == 2 else "")
                                      + ("" if w["backend"] is None else ", transport " + w["backend"])
                                      + (" (SELF-HALO TEST: the neighbour is this rank)" if args.self_halo else ""),
                       "step": "SSP-RK3 time step = 3 x (halo, limiter, halo, [dt], rhs, update)"},
            "roofline": {"bound": "hbm", "kernel": "qdg::k_rhs_p1w (tile / face-task RHS; stage 0: + CFL dt; stages 1,2: + fused RK update; 357 B/tet counted for every launch)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "avg_launch_ms": w["avg_ms"], "launches": w["launches"],
                         "algorithmic_bytes_per_launch": w["alg"],
                         "fp64_issue": valu_reading("p1", w["ntet_local"], w["avg_ms"])},
            # the three readings of "element update" -- `value` is the full-stage rate (limiter +
            # dt + RHS + RK update + halo: every tet through every RK stage, over the wall time of the step);
            # the RHS-only rate divides by the time of the RHS kernels alone; per time step = value / 3
            "rates": {"full_stage_M_per_s": w["ntet"] * 3 * args.steps / w["el"] / 1e6,
                      "rhs_only_M_per_s_rank0": w["ntet_local"] / (w["avg_ms"] * 1e-3) / 1e6,
                      "per_time_step_M_per_s": w["ntet"] * args.steps / w["el"] / 1e6,
                      # the same K steps timed right after the initial discontinuity (almost every tile unchanged
                      # by the limiter, nothing written back): the pre-round-5 headline
                      "cold_start_M_per_s": None if w["el_cold"] is None else w["ntet"] * 3 * args.steps / w["el_cold"] / 1e6,
                      # ... and with the limiter writing back every tile, changed or not: the rate of a flow that is
                      # developed everywhere, a lower bound that does not depend on the state (same results)
                      "limiter_writes_every_tile_M_per_s": w["ntet"] * 3 * args.steps / w["el_all"] / 1e6},
            # who ran: ranks in the RCCL communicator as RCCL counts them (ncclCommCount; None without RCCL),
            # and every rank's chunk, neighbours, device and own RHS launch time
            "ranks_seen_by_rccl": w["ranks_seen_by_rccl"], "per_rank": w["per_rank"],
            "dt_last": w["dt_last"],
            "check": {"mass_drift": float(w["drift"][0]), "energy_drift": float(w["drift"][1]),
                      "note": "relative change of total mass / total energy over the whole run "
                              "(conserved in this set-up; run aborts above 1e-9)"},
        }
        if ns is not None:
            a2, f2 = roof(ns)
            out["north_star_point"] = {
                "workload": "same physics, FIXED-SIZE Kuhn-tet box %d^3 hexes = %d tets cut %dx%dx%d across "
                            "the %d rank(s): BASELINE.json north_star (P1 RHS at ~10 M tets; strong scaling 1->8)"
                            % ((args.strong_nx, ns["ntet"]) + parts + (world,)),
                "scaling": "strong", "tets_total": ns["ntet"], "tets_rank0": ns["ntet_local"],
                "ranks_seen_by_rccl": ns["ranks_seen_by_rccl"], "per_rank": ns["per_rank"],
                "steps": ns["steps"], "warmup": ns["warmup"],
                "value": ns["ntet"] * 3 * ns["steps"] / ns["el"] / 1e6, "unit": "M element-updates/s",
                "ms_per_step": ns["el"] / ns["steps"] * 1e3,
                "developed_flow": {"untimed_steps_first": ns["develop"], "flow_time": ns["t_flow"]},
                "rates": {"full_stage_M_per_s": ns["ntet"] * 3 * ns["steps"] / ns["el"] / 1e6,
                          "rhs_only_M_per_s_rank0": ns["ntet_local"] / (ns["avg_ms"] * 1e-3) / 1e6,
                          "per_time_step_M_per_s": ns["ntet"] * ns["steps"] / ns["el"] / 1e6,
                          "cold_start_M_per_s": None if ns["el_cold"] is None else ns["ntet"] * 3 * ns["steps"] / ns["el_cold"] / 1e6,
                          "limiter_writes_every_tile_M_per_s": ns["ntet"] * 3 * ns["steps"] / ns["el_all"] / 1e6},
                "roofline": {"bound": "hbm", "kernel": "qdg::k_rhs_p1w on rank 0's chunk", "achieved": a2,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": f2, "avg_launch_ms": ns["avg_ms"],
                             "launches": ns["launches"], "algorithmic_bytes_per_launch": ns["alg"],
                             "traffic": None, "fp64_issue": valu_reading("p1", ns["ntet_local"], ns["avg_ms"])},
                "check": {"mass_drift": float(ns["drift"][0]), "energy_drift": float(ns["drift"][1])},
            }
        if world == 1 and not args.no_amr and not args.self_halo:
            out["amr_point"] = amr_point(local_rank)
            # config 5 at the north-star box: 119^3 x 6 = 10.1 M -> 80.9 M tets on one GPU
            # (two runs: cold, where the re-mesh pays the driver for every buffer itself, and with the
            # run's memory budget reserved from the driver at start-up -- what a run that knows it refines does)
            out["amr_point"]["at_north_star_size_cold"] = amr_point(local_rank, nx=args.strong_nx, steps=5,
                                                                    with_partition=False, reserve=False)
            out["amr_point"]["at_north_star_size"] = amr_point(local_rank, nx=args.strong_nx, steps=5,
                                                               with_partition=False, reserve=True)
            # ... and what one rank of a decomposition of that box spends: rank 0 of a 2x1x1 cut, 5.06 M owned tets ->
            # 40.4 M, alone on the GPU (round 4 kept both 40 M-tet chunks of the cut on the one device, so the second
            # chunk's re-mesh paid the driver for memory: not a rank's figure)
            out["amr_point"]["one_rank_of_a_decomposition_at_north_star_size"] = amr_one_rank(local_rank, args.strong_nx)
        if world == 1 and not args.no_config3 and not args.self_halo:
            out["config3_point"] = config3_point(local_rank, args.config3_nx)
        if world == 1 and not args.no_config4 and not args.self_halo:
            out["config4_point"] = config4_point(local_rank, develop=args.develop // 2)
        if world == 1 and not args.no_real_mesh and not args.self_halo:
            out["real_mesh_point"] = real_mesh_point(local_rank)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
