// qdg_handles.hpp -- the opaque handles of the C ABI (qdg_ctx, qdg_mesh) and the owning
// device buffer they are made of; shared by the translation units of libqdg that create or
// fill a mesh handle (qdg_api.cpp: upload from host arrays; qdg_devmesh.hip: layout built on
// the device).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/qdg.h"
#include "qdg_device.hpp"
#include "qdg_host.hpp"
#include "qdg_pool.hpp"

namespace qdg {

// owning device buffer
template <class T> struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) dev_free(p); }
  hipError_t alloc(size_t count)
  {
    if (p) { dev_free(p); p = nullptr; }
    n = count;
    if (count == 0) return hipSuccess;
    return dev_alloc((void**)&p, count * sizeof(T));
  }
  hipError_t upload(const std::vector<T>& h, hipStream_t s)
  {
    hipError_t e = alloc(h.size());
    if (e != hipSuccess || h.empty()) return e;
    e = hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(s);   // h may be a temporary of the caller
  }
};

}  // namespace qdg


namespace qdg {
// qdg_ctx_set_option: tuning / A-B switches (defaults = the product path)
struct Options {
  int p1_rhs = 0;        // DG-P1 RHS: 0 tile / face-task kernels (LDS atomics), 1 element-centric (bitwise reproducible)
  int fused_update = 1;  // stage-0 RK update fused with the Superbee limiter of stage 1
  int renumber = 1;      // Morton order of the interior tets (0: caller's order; layout experiments)
  int host_layout = 0;   // qdg_mesh_from_connectivity: 1 routes through qdg_mesh_upload's host code (A/B)
  int orient_by_gid = 1; // meshes built WITH global tet ids (qdg_mesh_*_gid): left tet of a face = lower global id
                         // (a partitioned run takes the serial run's HLLC branches); 0: chunk-local rule
  int keep_pool = 0;     // 1: qdg_ctx_destroy of the last context keeps the device buffer cache
  int keep_connectivity = 0;  // 1: device-built meshes without ghosts keep connectivity, coordinates, esuel and
                              // boundary faces (caller's numbering) resident: qdg_mesh_refine_uniform needs them
  int halo_depth = 1;    // 2: chunks built from now on are meant for two ghost layers (qdg_halo_set_depth): the tets within
                         // two faces of a ghost go last in the device order, so that every send row of the two-layer
                         // plan is a trailing row and the packs can be folded into the producing kernels
  int limiter_write_all = 0;  // 1: k_superbee writes back every tile (it normally skips tiles it leaves unchanged): the
                              // rate of a flow that is developed EVERYWHERE, whatever the state; measurement only
  int graph_step = 0;    // 1: qdg_step_comm replays its launch sequence (kernels + RCCL) as a hipGraph per
                         // buffer-rotation phase; falls back to plain launches where capture is refused
};
}  // namespace qdg

struct qdg_ctx {
  qdg_config cfg;
  qdg::Options opt;
  std::vector<int32_t> bc_sideset, bc_type;
  qdg::Phys ph;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  ~qdg_ctx()
  {
    if (own_stream && stream) {
      (void)hipSetDevice(device);
      (void)hipStreamSynchronize(stream);
      (void)hipStreamDestroy(stream);
    }
  }
};

struct qdg_mesh {
  qdg_ctx* ctx = nullptr;
  qdg::DevMesh dm{};
  int ndof = 1, nprop = 5;
  size_t nie = 0, ne = 0, stride = 0;
  // mesh
  qdg::DevBuf<int> inpoel, nbr, finfo, fid, d2h;
  qdg::DevBuf<double> x, y, z, farea, fnx, fny, fnz, vol, fgeo, xyz4;
  qdg::DevBuf<int> tile_row, tile_off, task_a, task_nb, task_f;
  qdg::DevBuf<double> tgeo;
  // fields (SoA planes [nprop][stride])
  qdg::DevBuf<double> U, Un, R, W;     // W: scratch state (stateless ops, WENO ping-pong)
  qdg::DevBuf<double> aos;             // [ne*nprop] staging in the caller's layout
  qdg::DevBuf<double> blockmin, dtraw, dtdev, diagpart, diagout;
  // the three field buffers U, Un, W rotate: Ucur = current state, Unp = the
  // stage-0 state of the running step (may alias Ucur until the first update),
  // the remaining one is free (RK output / WENO ping-pong)
  double* Ucur = nullptr;
  double* Unp = nullptr;
  double* Upending = nullptr;     // output of a fused RHS+RK launch, adopted by qdg_stage_update
  bool skip_ghost_carry = false;  // set by qdg_step_comm around an update whose ghost rows are received next
  double* carry_src = nullptr;    // buffer that still holds the ghost rows of a skipped carry (qdg_step_comm's error path)
  double* carry_pending = nullptr;  // after a qdg_step_comm: the buffer whose ghost rows belong to the current state;
                                    // copied over by the next entry point that is not a qdg_step_comm (which
                                    // starts by receiving every ghost row anyway)
  qdg::DevBuf<double> S1, S2;          // scratch of the stateless operators (allocated on first use)
  qdg::DevBuf<int> ndofel, ndofel2;    // p-adaptive DG: DG::m_ndof per device row (+ Jacobi copy)
  qdg::DevBuf<double> fout;            // field output staging (allocated on first use)
  // halo
  bool ghost_nbr = false;         // the nbr planes hold the ghost rows' neighbours too (device-built meshes)
  size_t nghost1 = 0;             // > 0: two ghost layers, the rank limits its nghost1 layer-1 ghosts itself
  size_t nnbr = 0, nsend = 0, nrecv = 0;
  std::vector<int32_t> nbr_rank;
  std::vector<size_t> send_off, recv_off;
  qdg::DevBuf<int> send_elem;
  qdg::DevBuf<int> fold_slot;     // [FOLD_SLOTS * (nie - ninner)] slab rows of every halo-adjacent row (-1 = none); empty: no folding
  const double* slab_ready_for = nullptr;   // the state whose send rows the slab already holds (qdg_step_comm)
  qdg::DevBuf<double> send_slab, recv_slab;
  double* send_ptr = nullptr;     // slabs in use (own or caller-provided)
  double* recv_ptr = nullptr;
  double* dt_ptr = nullptr;       // dt scalar in use
  size_t nnode_used = 0;
  // connectivity kept on the device for a re-mesh that does not go through the host (qdg_devmesh.hip)
  struct Keep;
  Keep* keep = nullptr;
  void (*keep_free)(Keep*) = nullptr;
  // measurement: event pairs around the RHS kernel (cont: second part of a split launch) and, inside
  // qdg_step_comm / qdg_halo_exchange / qdg_stage_dt_allreduce, around every exchange and all-reduce
  // (ev_kind: 0 RHS launch, 1 halo exchange = pack + grouped send / receive, 2 dt all-reduce)
  bool prof = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  std::vector<char> ev_cont, ev_kind;
  size_t ev_used = 0;
  // qdg_step_comm as a hipGraph (option graph_step): one executable graph per entry state
  struct StepGraph {
    const double* ucur_in; double t, tleft; bool slab_ready_in;
    hipGraphExec_t exec;
    double* ucur_out; const double* slab_ready_out; double* carry_out;
  };
  std::vector<StepGraph> step_graphs;
  int graph_state = 0;            // 0 not tried, 1 in use, -1 capture refused (graph_error says why)
  int graph_warm = 0;             // plain steps taken (RCCL sets its connections up in the first ones)
  long long graph_replays = 0;
  std::string graph_error;
  ~qdg_mesh()
  {
    for (auto& e : ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto& g : step_graphs) (void)hipGraphExecDestroy(g.exec);
    if (keep && keep_free) keep_free(keep);
  }
};


namespace qdg {
// qdg_api.cpp
int mesh_alloc_state(qdg_mesh* m, int ntile);
// ghost rows of the current state that a qdg_step_comm left in another buffer: copied over now (call before the
// state's rows are read or replaced outside the entry points of qdg_api.cpp)
int mesh_flush_carry(qdg_mesh* mesh);
// qdg_devmesh.hip: the halo plan of a chunk whose handle keeps its connectivity (for its re-mesh)
void keep_set_plan(qdg_mesh* m, size_t nnbr, const int32_t* nbr_rank, const size_t* recv_off);
// qdg_devmesh.hip: device side of qdg_state_transfer / qdg_state_migrate
int dev_state_transfer(qdg_mesh* from, qdg_mesh* to, const size_t* parent_of_child);
int dev_state_migrate(qdg_mesh* from, const size_t* from_gid, qdg_mesh* to, const size_t* to_gid, size_t* nmoved);
}  // namespace qdg
