"""Time loop of one DG mesh chunk on the device-resident path.

Mirror of the `DG` chare's per-stage sequence (src/Inciter/DG.cpp and dg.ci:57-70)

    next   : send own boundary-adjacent solution      (DG.cpp:1009-1039)  -> exchange()
    lim    : ghosts <- received, limiter, send limited (DG.cpp:1229-1282)  -> stage_limit(), exchange()
    dt     : ghosts <- received, CFL dt, min-reduce    (DG.cpp:1360-1430)  -> stage_rhs_dt(), allreduce_min
    solve  : Un=U (stage 0), rhs, SSP-RK3 update       (DG.cpp:1432-1508)  -> stage_rhs_dt(), stage_update()

with the Charm++ messages replaced by point-to-point sends between the ranks of
one node and the `contribute(min)` by an all-reduce of one double that stays on
the device.  The fields never leave HBM.  One process drives one GPU.

Transports:
  RcclComm   the product path on GPUs: libqdg's own RCCL calls (qdg_halo_exchange,
             qdg_stage_dt_allreduce; the whole step is ONE call, qdg_step_comm);
             torch.distributed only carries the 128-byte RCCL id at start-up
  TorchComm  torch.distributed point-to-point; with backend "gloo" the slabs are
             staged through the host (CPU test of the multi-rank logic, or several
             ranks sharing one GPU)
"""
import numpy as np


class SerialComm:
    """single chunk: nothing to exchange"""
    rank, size, backend = 0, 1, "none"

    def sendrecv(self, drv):
        pass

    def allreduce_min(self, drv):
        pass


class TorchComm:
    """torch.distributed transport.  backend "nccl" (= RCCL on ROCm): device
    slabs are sent directly, grouped per exchange; backend "gloo": slabs are
    staged through the host (CPU tests, or several ranks sharing one GPU)."""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self.backend = dist.get_backend()

    def sendrecv(self, drv):
        dist, torch = self.dist, self.torch
        ops, keep = [], []
        for i, q in enumerate(drv.nbr_rank):
            s0, s1 = drv.send_off[i] * drv.roww, drv.send_off[i + 1] * drv.roww
            r0, r1 = drv.recv_off[i] * drv.roww, drv.recv_off[i + 1] * drv.roww
            if self.backend == "nccl":
                sb, rb = drv.send_slab[s0:s1], drv.recv_slab[r0:r1]
            else:
                sb = drv.send_slab[s0:s1].cpu()
                rb = torch.empty(r1 - r0, dtype=torch.float64)
                keep.append((rb, r0, r1))
            ops.append(dist.P2POp(dist.isend, sb, q))
            ops.append(dist.P2POp(dist.irecv, rb, q))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for rb, r0, r1 in keep:
            drv.recv_slab[r0:r1].copy_(rb)

    def allreduce_min(self, drv):
        dist = self.dist
        if self.backend == "nccl":
            dist.all_reduce(drv.dt_buf, op=dist.ReduceOp.MIN)
        else:
            h = drv.dt_buf.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MIN)
            drv.dt_buf.copy_(h)


class SelfComm(TorchComm):
    """One rank whose halo neighbours are itself (send slab -> recv slab on the
    device).  Not a physical set-up: the reference transport against which the
    RCCL self send/recv is checked on a single GPU."""

    def __init__(self):
        import torch
        self.torch, self.dist = torch, None
        self.rank, self.size, self.backend = 0, 1, "self"

    def sendrecv(self, drv):
        drv.recv_slab.copy_(drv.send_slab)

    def allreduce_min(self, drv):
        pass


class RcclComm:
    """libqdg's RCCL transport.  `bootstrap` spreads rank 0's unique id: by
    default a torch.distributed broadcast (any backend)."""
    backend = "rccl"

    def __init__(self, ctx, rank=None, size=None, unique_id=None):
        from . import capi
        if unique_id is None:
            import torch
            import torch.distributed as dist
            rank, size = dist.get_rank(), dist.get_world_size()
            dev = torch.device("cuda", ctx.cfg.device) if dist.get_backend() == "nccl" else torch.device("cpu")
            t = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                t = torch.frombuffer(bytearray(capi.Comm.unique_id()), dtype=torch.uint8).to(dev)
            dist.broadcast(t, src=0)
            unique_id = bytes(t.cpu().numpy().tobytes())
        self.rank, self.size = rank, size
        self.comm = capi.Comm(ctx, size, rank, unique_id)

    def close(self):
        self.comm.close()


class DGDriver:
    def __init__(self, ctx, mesh, nbr_rank=(), send_lists=(), recv_counts=(), comm=None, nghost1=0):
        """nghost1 > 0: the chunk has two ghost layers (meshgen.kuhn_box_chunk / partition.build_chunk with
        depth = 2; nbr_rank / send_lists / recv_counts per (rank, layer) entry): the rank limits its nghost1
        layer-1 ghosts itself and the exchange of the limited solution (comlim) is dropped"""
        self.ctx, self.mesh = ctx, mesh
        self.deep = nghost1 > 0
        self.comm = comm or SerialComm()
        self.nprop = mesh.nprop
        self.pref = bool(ctx.cfg.pref)
        self.nbr_rank = list(nbr_rank)
        self.limiter_active = ctx.cfg.limiter != 0 and ctx.ndof > 1
        self.send_slab = self.recv_slab = self.dt_buf = None
        self.distributed = isinstance(self.comm, TorchComm)
        self.rccl = isinstance(self.comm, RcclComm)
        if self.rccl:
            mesh.halo_setup(self.nbr_rank, send_lists, recv_counts)
            if self.deep:
                mesh.halo_set_depth(nghost1)
        if self.distributed:
            torch = self.comm.torch
            dev = torch.device("cuda", torch.cuda.current_device())
            mesh.halo_setup(self.nbr_rank, send_lists, recv_counts)
            if self.deep:
                mesh.halo_set_depth(nghost1)
            self.send_off, self.recv_off = mesh.send_off, mesh.recv_off
            ns, nr = mesh.halo_sizes()
            # doubles per slab row (nprop, + 1 for the tet's ndof with p-adaptive DG)
            self.roww = mesh.halo_buffers()[2] // 8
            # torch owns the slabs and the dt scalar so that the communication
            # library sees ordinary device tensors; the kernels write into them
            self.send_slab = torch.zeros(max(1, ns * self.roww), dtype=torch.float64, device=dev)
            self.recv_slab = torch.zeros(max(1, nr * self.roww), dtype=torch.float64, device=dev)
            self.dt_buf = torch.zeros(1, dtype=torch.float64, device=dev)
            mesh.halo_use_buffers(self.send_slab.data_ptr(), self.recv_slab.data_ptr())
            mesh.stage_dt_use_buffer(self.dt_buf.data_ptr())
            # all kernels and all communication on torch's current stream
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def exchange(self):
        if not self.distributed or not self.nbr_rank:
            return
        self.mesh.halo_pack()
        self.comm.sendrecv(self)
        self.mesh.halo_unpack()

    def step(self, t, tleft=1e300):
        m = self.mesh
        if self.rccl:                            # the same sequence, inside libqdg
            m.step_comm(self.comm.comm, t, tleft)
            return
        if isinstance(self.comm, SerialComm) and not self.nbr_rank:
            m.step(t, tleft, want_dt=False)      # single chunk: qdg_step, one call per time step
            return
        for stage in range(3):
            if self.pref and stage == 0:
                m.stage_pdg_eval()               # DG::next: eval_ndof
            self.exchange()                      # comsol (rows + ndof)
            if self.pref and stage == 0:
                m.stage_pdg_propagate()          # DG::lim: propagate_ndof (+ zeroing of DG::solve)
            m.stage_limit()                      # (two ghost layers: also limits the layer-1 ghosts)
            if (self.limiter_active and not self.deep) or (self.pref and stage == 0):
                self.exchange()                  # comlim (a no-op copy without a limiter;
                                                 # with pdg it carries the propagated ndof)
            # rhs; at stage 0 it also yields the local dt (the reference computes
            # dt first, DG.cpp:1360-1430, from the same state; R does not depend on dt)
            m.stage_rhs_dt(stage, t, tleft)
            if stage == 0 and self.distributed:
                self.comm.allreduce_min(self)        # contribute(min), DG.cpp:1428-1429
            m.stage_update(stage)

    def dt_taken(self):
        return self.mesh.stage_dt_get()


class LocalChunks:
    """All chunks of a decomposition driven by ONE process on ONE GPU under ONE context (one
    stream): ghost rows move between the chunks' halo slabs as device copies (qdg_halo_copy),
    the time step is the minimum over the chunks (read back, like DG::dt's contribute(min)).
    Same per-stage sequence as DGDriver.step.  For tests of a decomposition (qdg_partition /
    qdg_chunk_build) on a one-GPU box; a production run has one process per GPU (DGDriver +
    RcclComm).  No torch here: a process that initialised HIP through libqdg must not bring up
    PyTorch's own HIP runtime afterwards."""

    def __init__(self, ctx, meshes, chunks):
        self.ctx, self.meshes, self.chunks = ctx, meshes, chunks
        self.pref = bool(ctx.cfg.pref)
        self.limiter_active = ctx.cfg.limiter != 0 and ctx.ndof > 1
        self.soff, self.roff = [], []
        # two ghost layers (chunks built with depth = 2): one plan entry per (rank, layer), no comlim exchange
        self.deep = all(ch.get("depth", 1) == 2 for ch in chunks) and len(chunks) > 1
        for mesh, ch in zip(meshes, chunks):
            mesh.halo_setup(ch["nbr_rank"], ch["send_lists"], ch["recv_counts"])
            if self.deep:
                mesh.halo_set_depth(ch["nghost1"])
            self.soff.append(np.concatenate([[0], np.cumsum([len(s) for s in ch["send_lists"]])]).astype(np.int64))
            self.roff.append(np.concatenate([[0], np.cumsum(ch["recv_counts"])]).astype(np.int64))

    def exchange(self):
        for m in self.meshes:
            m.halo_pack()
        for r, ch in enumerate(self.chunks):
            lay = ch.get("nbr_layer") or [1] * len(ch["nbr_rank"])
            for i, q in enumerate(ch["nbr_rank"]):
                oth = self.chunks[q]
                olay = oth.get("nbr_layer") or [1] * len(oth["nbr_rank"])
                j = [k for k, (qq, ll) in enumerate(zip(oth["nbr_rank"], olay)) if qq == r and ll == lay[i]][0]
                n = self.roff[r][i + 1] - self.roff[r][i]
                assert n == self.soff[q][j + 1] - self.soff[q][j]
                self.meshes[r].halo_copy_from(self.roff[r][i], self.meshes[q], self.soff[q][j], n)
        for m in self.meshes:
            m.halo_unpack()

    def step(self, t, tleft=1e300):
        for stage in range(3):
            if self.pref and stage == 0:
                for m in self.meshes:
                    m.stage_pdg_eval()
            self.exchange()
            if self.pref and stage == 0:
                for m in self.meshes:
                    m.stage_pdg_propagate()
            for m in self.meshes:
                m.stage_limit()
            if (self.limiter_active and not self.deep) or (self.pref and stage == 0):
                self.exchange()
            for m in self.meshes:
                m.stage_rhs_dt(stage, t, tleft)
            if stage == 0:
                dt = min(m.stage_dt_get() for m in self.meshes)
                for m in self.meshes:
                    m.stage_dt_set(dt)
            for m in self.meshes:
                m.stage_update(stage)
        return dt
