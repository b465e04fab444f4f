// test_level2_driver.cpp -- integration Level 2 driven from C++ through the C ABI
// alone: what a DG chare array would call once the fields live on the device.  No Python and no
// adapter class between the calls:
//
//   qdg_partition -> qdg_chunk_build (per rank) -> qdg_mesh_from_chunk -> qdg_halo_setup ->
//   qdg_state_initialize ->
//   N x SSP-RK3 step in the DG chare's order (src/Inciter/dg.ci:57-70: next -> comsol -> lim ->
//        comlim -> dt -> solve): halo pack / transport / unpack, qdg_stage_limit, halo again,
//        qdg_stage_rhs_dt, min of the chunks' dt (contribute(min), DG.cpp:1428-1429),
//        qdg_stage_update ->
//   re-mesh on the decomposition (DG::resizePostAMR, DG.cpp:1536-1612): qdg_refine_chunk ->
//        qdg_mesh_from_chunk -> qdg_halo_setup -> qdg_state_transfer ->
//   M more steps -> states out with their global tet ids.
//
// The chunks of the decomposition all sit on this process's one GPU, so the transport between
// them is qdg_halo_copy (send slab of the sender -> receive slab of the receiver); a run with one
// process per GPU replaces exactly that call (and the dt minimum) by qdg_halo_exchange /
// qdg_stage_dt_allreduce or its own messages.  tests/test_gpu_cpp_adapter.py writes the mesh and
// compares the result with the oracle's serial run across the same refinement.
//
//   usage: test_level2_driver mesh.bin out.bin nparts nsteps_before nsteps_after
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "qdg.h"

static void check(int rc, const char* what)
{
  if (rc != 0) { fprintf(stderr, "%s: %s\n", what, qdg_last_error()); exit(1); }
}
#define CHECK(call) check((call), #call)

template <class T> static std::vector<T> rd(FILE* f, size_t n)
{
  std::vector<T> v(n);
  if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
  return v;
}

// one rank's chunk as the host side of a DG chare holds it
struct Chunk {
  size_t nielem = 0, nunk = 0, nnode = 0;
  std::vector<size_t> inpoel, gid;                 // local node ids; global tet ids
  std::vector<double> x, y, z;
  std::vector<size_t> tri; std::vector<int32_t> tri_set;     // side-set triangles, local node ids
  std::vector<int32_t> nbr_rank, nbr_layer;        // one plan entry per (neighbour rank, ghost layer)
  size_t nghost1 = 0;                              // layer-1 ghosts (= all ghosts with one layer)
  std::vector<size_t> send_off, send_elem, recv_off;
  qdg_mesh* mesh = nullptr;
};

static bool g_device_remesh = false;    // 7th argument "device": the re-mesh by qdg_mesh_refine_chunk
static int g_depth = 1;                 // 7th argument "deep": chunks with TWO ghost layers (qdg_chunk_build_depth,
                                        // qdg_halo_set_depth, qdg_refine_chunk_depth): every rank limits its layer-1
                                        // ghosts itself and the exchange of the limited solution (comlim) is dropped

static void to_device(qdg_ctx* ctx, Chunk& c)
{
  // (with the tets' global ids: faces oriented as in the serial run, and what the device re-mesh orders by)
  CHECK(qdg_mesh_from_chunk_gid(ctx, c.nielem, c.nunk, c.nnode, c.inpoel.data(), c.x.data(), c.y.data(), c.z.data(),
                                c.tri_set.size(), c.tri.data(), c.tri_set.data(), c.gid.data(), &c.mesh));
  CHECK(qdg_halo_setup(c.mesh, c.nbr_rank.size(), c.nbr_rank.data(), c.send_off.data(), c.send_elem.data(),
                       c.recv_off.data()));
  if (g_depth == 2) CHECK(qdg_halo_set_depth(c.mesh, c.nghost1));
}

// DG::next -> comsol / DG::lim -> comlim: every chunk packs, rows travel, every chunk unpacks
static void exchange(std::vector<Chunk>& ch)
{
  for (auto& c : ch) CHECK(qdg_halo_pack(c.mesh));
  for (size_t r = 0; r < ch.size(); ++r)
    for (size_t i = 0; i < ch[r].nbr_rank.size(); ++i) {
      Chunk& q = ch[(size_t)ch[r].nbr_rank[i]];
      size_t j = 0;                    // the other side's entry for (this rank, the same layer)
      while (j < q.nbr_rank.size() && !(q.nbr_rank[j] == (int32_t)r && q.nbr_layer[j] == ch[r].nbr_layer[i])) ++j;
      const size_t n = ch[r].recv_off[i + 1] - ch[r].recv_off[i];
      if (j >= q.nbr_rank.size() || n != q.send_off[j + 1] - q.send_off[j]) { fprintf(stderr, "halo plan mismatch\n"); exit(1); }
      CHECK(qdg_halo_copy(ch[r].mesh, ch[r].recv_off[i], q.mesh, q.send_off[j], n));
    }
  for (auto& c : ch) CHECK(qdg_halo_unpack(c.mesh));
}

// one SSP-RK3 time step of all chunks, stage by stage in the order of dg.ci:57-70
static double step(std::vector<Chunk>& ch, double t)
{
  double dt = 0.0;
  for (int stage = 0; stage < 3; ++stage) {
    exchange(ch);                                              // next -> comsol
    for (auto& c : ch) CHECK(qdg_stage_limit(c.mesh));         // lim (two layers: the layer-1 ghosts too)
    if (g_depth == 1) exchange(ch);                            // -> comlim
    for (auto& c : ch) CHECK(qdg_stage_rhs_dt(c.mesh, stage, t, 1e300));    // dt (stage 0), solve: rhs
    if (stage == 0) {                                          // contribute(min)
      dt = 1e300;
      for (auto& c : ch) { double d; CHECK(qdg_stage_dt_get(c.mesh, &d)); dt = std::min(dt, d); }
      for (auto& c : ch) CHECK(qdg_stage_dt_set(c.mesh, dt));
    }
    for (auto& c : ch) CHECK(qdg_stage_update(c.mesh, stage)); // solve: SSP-RK3 update
  }
  return dt;
}

int main(int argc, char** argv)
{
  if (argc != 6 && argc != 7) { fprintf(stderr, "usage: %s mesh.bin out.bin nparts nsteps_before nsteps_after [device|deep]\n", argv[0]); return 2; }
  g_device_remesh = argc == 7 && std::string(argv[6]) == "device";
  g_depth = (argc == 7 && std::string(argv[6]) == "deep") ? 2 : 1;
  const int nparts = atoi(argv[3]), n0 = atoi(argv[4]), n1 = atoi(argv[5]);
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("mesh"); return 2; }
  auto hdr = rd<uint64_t>(f, 3);                       // nnode, nelem, ntri
  const size_t nnode = hdr[0], nelem = hdr[1], ntri = hdr[2];
  std::vector<double> xyz[3];
  for (int d = 0; d < 3; ++d) xyz[d] = rd<double>(f, nnode);
  std::vector<size_t> inpoel; { auto v = rd<uint64_t>(f, 4 * nelem); inpoel.assign(v.begin(), v.end()); }
  std::vector<size_t> tri; { auto v = rd<uint64_t>(f, 3 * ntri); tri.assign(v.begin(), v.end()); }
  auto triset = rd<int32_t>(f, ntri);
  fclose(f);

  // the run's configuration: Sod shock tube, DG-P1, HLLC, Superbee, CFL 0.3 (BASELINE config 2 / 5)
  const int32_t bcset[6] = { 1, 2, 3, 4, 5, 6 };
  const int32_t bctype[6] = { QDG_BC_EXTRAPOLATE, QDG_BC_EXTRAPOLATE, QDG_BC_SYMMETRY, QDG_BC_SYMMETRY,
                              QDG_BC_SYMMETRY, QDG_BC_SYMMETRY };
  qdg_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = (int32_t)sizeof cfg;
  cfg.ndof = cfg.rdof = 4; cfg.flux = QDG_FLUX_HLLC; cfg.limiter = QDG_LIMITER_SUPERBEEP1;
  cfg.problem = QDG_PROBLEM_SOD_SHOCKTUBE; cfg.nbc = 6; cfg.bc_sideset = bcset; cfg.bc_type = bctype;
  cfg.gamma = 1.4; cfg.cv = 717.5; cfg.cweight = 1.0; cfg.cfl = 0.3; cfg.pde = QDG_PDE_COMPFLOW; cfg.tolref = 0.1;
  qdg_ctx* ctx = nullptr;
  CHECK(qdg_ctx_create(&cfg, &ctx));
  if (g_device_remesh) CHECK(qdg_ctx_set_option(ctx, "keep_connectivity", 1));
  if (g_depth == 2) CHECK(qdg_ctx_set_option(ctx, "halo_depth", 2));

  // ---- decomposition (Partitioner + the DG chare's ghost set-up) ---------------------------
  std::vector<int32_t> part(nelem);
  CHECK(qdg_partition(nelem, inpoel.data(), nnode, xyz[0].data(), xyz[1].data(), xyz[2].data(), nparts, QDG_PART_RCB,
                      part.data()));
  std::vector<Chunk> ch((size_t)nparts);
  for (int r = 0; r < nparts; ++r) {
    Chunk& c = ch[(size_t)r];
    qdg_chunk* h = nullptr;
    CHECK(qdg_chunk_build_depth(nelem, nnode, inpoel.data(), nullptr, part.data(), nparts, r, g_depth, &h));
    size_t nnbr = 0, nsend = 0;
    CHECK(qdg_chunk_sizes(h, &c.nielem, &c.nunk, &c.nnode, &nnbr, &nsend));
    c.inpoel.resize(4 * c.nunk); c.gid.resize(c.nunk);
    std::vector<size_t> node_gid(c.nnode);
    c.nbr_rank.resize(nnbr); c.nbr_layer.resize(nnbr); c.send_off.resize(nnbr + 1); c.send_elem.resize(nsend); c.recv_off.resize(nnbr + 1);
    CHECK(qdg_chunk_layers(h, nullptr, &c.nghost1, c.nbr_layer.data()));
    CHECK(qdg_chunk_get(h, c.inpoel.data(), c.gid.data(), node_gid.data(), c.nbr_rank.data(), c.send_off.data(),
                        c.send_elem.data(), c.recv_off.data()));
    CHECK(qdg_chunk_destroy(h));
    c.x.resize(c.nnode); c.y.resize(c.nnode); c.z.resize(c.nnode);
    std::vector<long> g2l(nnode, -1);
    for (size_t n = 0; n < c.nnode; ++n) {
      g2l[node_gid[n]] = (long)n;
      c.x[n] = xyz[0][node_gid[n]]; c.y[n] = xyz[1][node_gid[n]]; c.z[n] = xyz[2][node_gid[n]];
    }
    // side-set triangles whose nodes are all in the chunk (the device mesh build matches them
    // with the faces of the OWNED tets and ignores the rest)
    for (size_t t = 0; t < ntri; ++t) {
      const long a = g2l[tri[3 * t]], b = g2l[tri[3 * t + 1]], d = g2l[tri[3 * t + 2]];
      if (a >= 0 && b >= 0 && d >= 0) {
        c.tri.push_back((size_t)a); c.tri.push_back((size_t)b); c.tri.push_back((size_t)d);
        c.tri_set.push_back(triset[t]);
      }
    }
    to_device(ctx, c);
    CHECK(qdg_state_initialize(c.mesh, 0.0));
  }

  double t = 0.0;
  std::vector<double> dts;
  for (int s = 0; s < n0; ++s) { const double dt = step(ch, t); t += dt; dts.push_back(dt); }

  // ---- re-mesh on the decomposition: every rank refines its own chunk ------------------------
  for (auto& c : ch) {
    std::vector<size_t> recv_counts(c.nbr_rank.size());
    for (size_t i = 0; i < recv_counts.size(); ++i) recv_counts[i] = c.recv_off[i + 1] - c.recv_off[i];
    if (g_device_remesh) {
      // the whole re-mesh of the rank's chunk on the device; the host keeps global ids and the new plan
      qdg_mesh* nm = nullptr; qdg_chunk_refined* h = nullptr;
      CHECK(qdg_mesh_refine_chunk(c.mesh, &nm, &h, 0));
      Chunk n;
      size_t ntri2 = 0, nsend = 0;
      CHECK(qdg_chunk_refined_sizes(h, &n.nielem, &n.nunk, &n.nnode, &ntri2, &nsend));
      n.gid.resize(n.nunk);
      n.nbr_rank = c.nbr_rank; n.nbr_layer = c.nbr_layer;
      n.send_off.resize(n.nbr_rank.size() + 1); n.send_elem.resize(nsend);
      CHECK(qdg_chunk_refined_get(h, nullptr, n.gid.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                  n.send_off.data(), n.send_elem.data(), recv_counts.data()));
      CHECK(qdg_chunk_refined_destroy(h));
      n.recv_off.assign(n.nbr_rank.size() + 1, 0);
      for (size_t i = 0; i < recv_counts.size(); ++i) n.recv_off[i + 1] = n.recv_off[i] + recv_counts[i];
      n.mesh = nm;
      CHECK(qdg_mesh_destroy(c.mesh));
      c = std::move(n);
      continue;
    }
    qdg_chunk_refined* h = nullptr;
    CHECK(qdg_refine_chunk_depth(c.nielem, c.nunk, c.nnode, c.inpoel.data(), c.x.data(), c.y.data(), c.z.data(),
                                 c.gid.data(), c.tri_set.size(), c.tri.data(), c.tri_set.data(), c.nbr_rank.size(),
                                 c.nbr_rank.data(), recv_counts.data(), g_depth, &h));
    Chunk n;
    size_t ntri2 = 0, nsend = 0;
    CHECK(qdg_chunk_refined_sizes(h, &n.nielem, &n.nunk, &n.nnode, &ntri2, &nsend));
    // (the refined chunk's plan entries are its own: with two layers they can differ from the old ones)
    size_t nent2 = 0;
    CHECK(qdg_chunk_refined_plan(h, &nent2, &n.nghost1, nullptr, nullptr));
    n.nbr_rank.resize(nent2); n.nbr_layer.resize(nent2);
    CHECK(qdg_chunk_refined_plan(h, nullptr, nullptr, n.nbr_rank.data(), n.nbr_layer.data()));
    recv_counts.assign(nent2, 0);
    n.inpoel.resize(4 * n.nunk); n.gid.resize(n.nunk);
    std::vector<size_t> parent(n.nunk);
    n.x.resize(n.nnode); n.y.resize(n.nnode); n.z.resize(n.nnode);
    n.tri.resize(3 * ntri2); n.tri_set.resize(ntri2);
    n.send_off.resize(n.nbr_rank.size() + 1); n.send_elem.resize(nsend);
    CHECK(qdg_chunk_refined_get(h, n.inpoel.data(), n.gid.data(), parent.data(), n.x.data(), n.y.data(), n.z.data(),
                                n.tri.data(), n.tri_set.data(), n.send_off.data(), n.send_elem.data(),
                                recv_counts.data()));
    CHECK(qdg_chunk_refined_destroy(h));
    n.recv_off.assign(n.nbr_rank.size() + 1, 0);
    for (size_t i = 0; i < recv_counts.size(); ++i) n.recv_off[i + 1] = n.recv_off[i] + recv_counts[i];
    to_device(ctx, n);
    CHECK(qdg_state_transfer(c.mesh, n.mesh, parent.data()));      // child <- parent, on the device
    CHECK(qdg_mesh_destroy(c.mesh));
    c = std::move(n);
  }
  for (int s = 0; s < n1; ++s) { const double dt = step(ch, t); t += dt; dts.push_back(dt); }

  // ---- results: time, the steps taken, per chunk the owned rows with their global ids -------
  FILE* o = fopen(argv[2], "wb");
  if (!o) { perror("out"); return 2; }
  const uint64_t nd = dts.size(), np = ch.size();
  fwrite(&t, 8, 1, o); fwrite(&nd, 8, 1, o); fwrite(dts.data(), 8, nd, o); fwrite(&np, 8, 1, o);
  for (auto& c : ch) {
    std::vector<double> U(c.nunk * 20);
    CHECK(qdg_state_download(c.mesh, U.data()));
    const uint64_t nie = c.nielem;
    fwrite(&nie, 8, 1, o);
    std::vector<uint64_t> g(c.gid.begin(), c.gid.begin() + (long)c.nielem);
    fwrite(g.data(), 8, nie, o);
    fwrite(U.data(), 8, nie * 20, o);
    CHECK(qdg_mesh_destroy(c.mesh));
  }
  fclose(o);
  CHECK(qdg_ctx_destroy(ctx));
  return 0;
}
