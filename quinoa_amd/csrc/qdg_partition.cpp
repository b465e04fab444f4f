// qdg_partition.cpp -- host side of the mesh decomposition for one-process-per-GPU
// runs: a geometric partitioner for an arbitrary tetrahedron connectivity and the
// builder of one rank's chunk (owned tets, one-layer face-neighbour ghosts, halo
// plan) in the data model the DG chare holds after its ghost set-up.
//
//   reference                                            here
//   ---------------------------------------------------  -------------------------
//   Partitioner::partition -> tk::geomPartMesh (Zoltan2   qdg_partition: the same kind
//   RCB / RIB / HSFC / MJ on the element centroids,      of cut (coordinate bisection or a
//   src/Inciter/Partitioner.cpp:137-170,                  space-filling-curve order of the
//   src/LoadBalance/ZoltanInterOp.cpp)                    centroids), own implementation
//   DG::DG ... DG::adj: chare-boundary faces, ghost       qdg_chunk_build: ghosts = tets of
//   tets and their numbering, m_ghostData / m_ghost       other ranks that share a FACE with
//   (src/Inciter/DG.cpp:134-949, 468-712)                 an owned tet, grouped by owner
//
// Zoltan2 itself is a third-party library that is absent here; WHICH tet lands in
// which part therefore differs from a reference run (as it does between the
// reference's own partitioners) -- the physics does not depend on it.
#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/qdg.h"
#include "qdg_host.hpp"

using namespace qdg;

namespace {

inline uint64_t spread21(uint64_t v)
{
  v &= 0x1fffff;
  v = (v | v << 32) & 0x1f00000000ffffULL;
  v = (v | v << 16) & 0x1f0000ff0000ffULL;
  v = (v | v << 8) & 0x100f00f00f00f00fULL;
  v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
  v = (v | v << 2) & 0x1249249249249249ULL;
  return v;
}

// recursive coordinate bisection of idx[b,e) into parts [p0, p0+np): split the longest
// extent at the weighted median (np/2 : np - np/2), so any number of parts works
void rcb(std::vector<int>& idx, size_t b, size_t e, int p0, int np, const std::vector<double>& cx,
         const std::vector<double>& cy, const std::vector<double>& cz, int32_t* part)
{
  if (np <= 1 || e - b <= 1) {
    for (size_t i = b; i < e; ++i) part[idx[i]] = p0;
    return;
  }
  double lo[3] = { DBL_MAX, DBL_MAX, DBL_MAX }, hi[3] = { -DBL_MAX, -DBL_MAX, -DBL_MAX };
  for (size_t i = b; i < e; ++i) {
    const double c[3] = { cx[idx[i]], cy[idx[i]], cz[idx[i]] };
    for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], c[d]); hi[d] = std::max(hi[d], c[d]); }
  }
  int ax = 0;
  for (int d = 1; d < 3; ++d) if (hi[d] - lo[d] > hi[ax] - lo[ax]) ax = d;
  const std::vector<double>& c = ax == 0 ? cx : ax == 1 ? cy : cz;
  const int npl = np / 2;
  const size_t mid = b + (size_t)((double)(e - b) * npl / np + 0.5);
  // ties broken by element id: the cut is a function of the mesh alone
  std::nth_element(idx.begin() + b, idx.begin() + mid, idx.begin() + e,
                   [&](int p, int q) { return c[p] < c[q] || (c[p] == c[q] && p < q); });
  rcb(idx, b, mid, p0, npl, cx, cy, cz, part);
  rcb(idx, mid, e, p0 + npl, np - npl, cx, cy, cz, part);
}

}  // namespace

extern "C" int qdg_partition(size_t nelem, const size_t* inpoel, size_t nnode, const double* x,
                             const double* y, const double* z, int nparts, int method, int32_t* part)
{
  QDG_TRY
  if (!inpoel || !x || !y || !z || !part) return fail("qdg_partition: null argument");
  if (nparts < 1) return fail("qdg_partition: nparts must be >= 1");
  if (nelem > (size_t)INT32_MAX) return fail("qdg_partition: too many elements");
  if (method != QDG_PART_RCB && method != QDG_PART_MORTON) return fail("qdg_partition: unknown method");
  std::vector<double> cx(nelem), cy(nelem), cz(nelem);
  for (size_t e = 0; e < nelem; ++e) {
    double s[3] = { 0.0, 0.0, 0.0 };
    for (int i = 0; i < 4; ++i) {
      const size_t n = inpoel[4 * e + i];
      if (n >= nnode) return fail("qdg_partition: inpoel entry out of range");
      s[0] += x[n]; s[1] += y[n]; s[2] += z[n];
    }
    cx[e] = 0.25 * s[0]; cy[e] = 0.25 * s[1]; cz[e] = 0.25 * s[2];
  }
  std::vector<int> idx(nelem);
  std::iota(idx.begin(), idx.end(), 0);
  if (method == QDG_PART_RCB) {
    rcb(idx, 0, nelem, 0, nparts, cx, cy, cz, part);
    return 0;
  }
  // space-filling curve: Morton order of the centroids, cut into nparts equal runs
  double lo[3] = { DBL_MAX, DBL_MAX, DBL_MAX }, hi[3] = { -DBL_MAX, -DBL_MAX, -DBL_MAX };
  for (size_t e = 0; e < nelem; ++e) {
    lo[0] = std::min(lo[0], cx[e]); hi[0] = std::max(hi[0], cx[e]);
    lo[1] = std::min(lo[1], cy[e]); hi[1] = std::max(hi[1], cy[e]);
    lo[2] = std::min(lo[2], cz[e]); hi[2] = std::max(hi[2], cz[e]);
  }
  const double ext = std::max({ hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], 1e-300 });
  std::vector<uint64_t> key(nelem);
  for (size_t e = 0; e < nelem; ++e) {
    const double c[3] = { cx[e], cy[e], cz[e] };
    uint64_t k = 0;
    for (int d = 0; d < 3; ++d) {
      const double t = (c[d] - lo[d]) / ext;
      k |= spread21((uint64_t)std::min(2097151.0, std::max(0.0, t * 2097152.0))) << d;
    }
    key[e] = k;
  }
  std::sort(idx.begin(), idx.end(), [&](int p, int q) { return key[p] < key[q] || (key[p] == key[q] && p < q); });
  for (size_t i = 0; i < nelem; ++i) part[idx[i]] = (int32_t)((i * (size_t)nparts) / nelem);
  return 0;
  QDG_CATCH
}

// ---------------------------------------------------------------------------------------
// Uniform 8:1 derefinement: the inverse of qdg_refine_uniform, for a mesh that IS a uniform refinement in this
// library's order (children 8 e + k of parent e, refine_one_to_eight's child list: (A,AB,AC,AD), (B,BC,AB,BD),
// (C,AC,BC,CD), (D,AD,CD,BD) and the four tets of the inner octahedron; old nodes before the edge midpoints; child
// triangles 4 t + k of boundary triangle t).  What the reference's Refiner does for `uniform_derefine` at t0
// (src/Inciter/Refiner.cpp:395-408 -> AMR::mesh_adapter_t::uniform_derefinement, AMR/mesh_adapter.cpp; parents restored
// from their children, AMR/refinement.hpp:726-800): its t0ref goldens of uniform -> uniform_derefine -> uniform hold
// the original mesh again after the derefinement step (tests/golden/t0ref_gauss_hump_udu.npz).  The parent is
// (A, B, C, D) = the first node of children 0-3; the structure of all eight children is verified.
extern "C" int qdg_derefine_uniform(size_t nelem, size_t nnode, const size_t* inpoel, const double* x, const double* y,
                                    const double* z, size_t ntri, const size_t* tri, qdg_refined** out)
{
  QDG_TRY
  if (!inpoel || !x || !y || !z || !out || (ntri && !tri)) return fail("qdg_derefine_uniform: null argument");
  *out = nullptr;
  if (nelem == 0 || nelem % 8 != 0) return fail("qdg_derefine_uniform: the number of tets is not a multiple of 8");
  if (ntri % 4 != 0) return fail("qdg_derefine_uniform: the number of boundary triangles is not a multiple of 4");
  const size_t np = nelem / 8;
  std::unique_ptr<qdg_refined> r(new qdg_refined);
  r->inpoel.resize(4 * np); r->parent.resize(np);
  size_t ncoarse = 0;
  for (size_t p = 0; p < np; ++p) {
    const size_t* c = inpoel + 32 * p;
    for (size_t i = 0; i < 32; ++i) if (c[i] >= nnode) return fail("qdg_derefine_uniform: inpoel entry out of range");
    const size_t A = c[0], B = c[4], C = c[8], D = c[12];
    const size_t AB = c[1], AC = c[2], AD = c[3], BC = c[5], BD = c[7], CD = c[11];
    const size_t want[8][4] = { { A, AB, AC, AD }, { B, BC, AB, BD }, { C, AC, BC, CD }, { D, AD, CD, BD },
                                { BC, CD, AC, BD }, { AB, BD, AC, AD }, { AB, BC, AC, BD }, { AC, BD, CD, AD } };
    for (int k = 0; k < 8; ++k)
      for (int i = 0; i < 4; ++i)
        if (c[4 * k + i] != want[k][i])
          return fail("qdg_derefine_uniform: tets " + std::to_string(8 * p) + "... are not the eight children of one tet "
                      "in the order of qdg_refine_uniform");
    r->inpoel[4 * p] = A; r->inpoel[4 * p + 1] = B; r->inpoel[4 * p + 2] = C; r->inpoel[4 * p + 3] = D;
    r->parent[p] = 8 * p;                       // (its first child: where a row copy takes the parent's state from)
    ncoarse = std::max({ ncoarse, A + 1, B + 1, C + 1, D + 1 });
  }
  // the old nodes come first: every midpoint id lies behind every corner id
  for (size_t p = 0; p < np; ++p) {
    const size_t* c = inpoel + 32 * p;
    const size_t mid[6] = { c[1], c[2], c[3], c[5], c[7], c[11] };
    for (size_t m : mid) if (m < ncoarse) return fail("qdg_derefine_uniform: a midpoint node is numbered before a corner node");
  }
  r->nnode = ncoarse;
  r->x.assign(x, x + ncoarse); r->y.assign(y, y + ncoarse); r->z.assign(z, z + ncoarse);
  r->tri.resize(3 * (ntri / 4));
  for (size_t t = 0; t < ntri / 4; ++t) {
    const size_t* q = tri + 12 * t;             // (a,ab,ac), (b,bc,ab), (c,ac,bc), (ab,bc,ac)
    if (q[1] != q[5] || q[2] != q[7] || q[4] != q[8] || q[9] != q[1] || q[10] != q[4] || q[11] != q[2])
      return fail("qdg_derefine_uniform: boundary triangles " + std::to_string(4 * t) + "... are not the four children of one triangle");
    r->tri[3 * t] = q[0]; r->tri[3 * t + 1] = q[3]; r->tri[3 * t + 2] = q[6];
    if (q[0] >= ncoarse || q[3] >= ncoarse || q[6] >= ncoarse) return fail("qdg_derefine_uniform: boundary triangle corner is a midpoint");
  }
  *out = r.release();
  return 0;
  QDG_CATCH
}

// ---------------------------------------------------------------------------------------
// Ghost layers and halo plan of ONE rank from the face adjacency of a mesh (or of the part of a mesh around
// the rank's tets), the owner rank of every tet and the tets' global ids.
//
//   layer 1 = tets of other ranks that share a face with an owned tet: what the DG chare keeps as ghosts
//             (src/Inciter/DG.cpp:468-712);
//   layer 2 (depth 2) = tets of other ranks, not in layer 1, that share a face with a layer-1 tet.  With it a
//             rank holds every input of the limiter of its layer-1 ghosts (Limiter.cpp:29-316 read the face
//             neighbours of a tet) and limits them ITSELF: the second exchange of every RK stage -- the limited
//             solution, DG::lim -> comlim, DG.cpp:1262-1282 -- is not needed, a step has 3 exchanges instead of 6.
//
// A plan ENTRY is a (neighbour rank, layer) pair: the layer-1 entries by ascending rank, then the layer-2
// entries by ascending rank; entry i has a send list (the owned tets in that rank's layer of THAT rank's plan,
// ordered by global id) and a receive range of the ghost rows (that rank's tets in this rank's layer, ordered
// by global id).  Both ranks of a pair derive matching lists without talking to each other.  Rank q's layer 2
// holds my tet t iff t is not in q's layer 1 and some face neighbour g of t (mine or a third rank's) is: i.e.
// owner(g) != q and g shares a face with a tet of q -- all within two faces of t, so the adjacency only has to
// be complete that far around the owned tets.
namespace {
struct GhostPlan {
  size_t nghost1 = 0;
  std::vector<size_t> ghost;                     // tets of the layers: layer 1 (by owner, gid), then layer 2
  std::vector<int32_t> entry_rank, entry_layer;
  std::vector<size_t> recv_off, send_off, send_elem;
};

int ghost_plan(size_t nelem, const int* esuel, const int32_t* owner, const size_t* gid, int rank, int depth,
               GhostPlan& gp, const char* who)
{
  auto gidof = [&](size_t e) { return gid ? gid[e] : e; };
  std::vector<char> layer(nelem, 0);
  using Pair = std::pair<int32_t, size_t>;        // (rank, tet)
  std::vector<Pair> recv[2], send[2];
  for (size_t e = 0; e < nelem; ++e) {
    if (owner[e] != rank) continue;
    for (int lf = 0; lf < 4; ++lf) {
      const int nb = esuel[4 * e + lf];
      if (nb < -1 || (nb >= 0 && (size_t)nb >= nelem)) return fail(std::string(who) + ": esuel entry out of range");
      if (nb >= 0 && owner[nb] != rank) {
        if (!layer[nb]) { layer[nb] = 1; recv[0].emplace_back(owner[nb], (size_t)nb); }
        send[0].emplace_back(owner[nb], e);
      }
    }
  }
  if (depth >= 2) {
    const size_t n1 = recv[0].size();
    for (size_t i = 0; i < n1; ++i) {
      const size_t g = recv[0][i].second;
      for (int lf = 0; lf < 4; ++lf) {
        const int nb = esuel[4 * g + lf];
        if (nb >= 0 && owner[nb] != rank && !layer[nb]) { layer[nb] = 2; recv[1].emplace_back(owner[nb], (size_t)nb); }
      }
    }
    // my tets in the layer 2 of rank q: not next to a tet of q, but next to a tet g (not q's) that is
    for (size_t e = 0; e < nelem; ++e) {
      if (owner[e] != rank) continue;
      int32_t direct[4]; int nd = 0;
      for (int lf = 0; lf < 4; ++lf) {
        const int nb = esuel[4 * e + lf];
        if (nb >= 0 && owner[nb] != rank) direct[nd++] = owner[nb];
      }
      for (int lf = 0; lf < 4; ++lf) {
        const int g = esuel[4 * e + lf];
        if (g < 0) continue;
        for (int l2 = 0; l2 < 4; ++l2) {
          const int a = esuel[4 * (size_t)g + l2];
          if (a < 0) continue;
          const int32_t q = owner[a];
          if (q == rank || q == owner[g]) continue;          // g must be in q's layer 1: not q's own tet
          bool isdirect = false;
          for (int i = 0; i < nd; ++i) isdirect = isdirect || direct[i] == q;
          if (!isdirect) send[1].emplace_back(q, e);
        }
      }
    }
  }
  auto order = [&](std::vector<Pair>& v) {
    std::sort(v.begin(), v.end(), [&](const Pair& p, const Pair& q) {
      return p.first != q.first ? p.first < q.first : gidof(p.second) < gidof(q.second); });
    v.erase(std::unique(v.begin(), v.end()), v.end());
  };
  gp.recv_off.assign(1, 0); gp.send_off.assign(1, 0);
  for (int l = 0; l < (depth >= 2 ? 2 : 1); ++l) {
    order(recv[l]); order(send[l]);
    std::vector<int32_t> ranks;
    for (const Pair& p : recv[l]) ranks.push_back(p.first);
    for (const Pair& p : send[l]) ranks.push_back(p.first);
    std::sort(ranks.begin(), ranks.end());
    ranks.erase(std::unique(ranks.begin(), ranks.end()), ranks.end());
    size_t i = 0, j = 0;
    for (int32_t q : ranks) {
      gp.entry_rank.push_back(q); gp.entry_layer.push_back(l + 1);
      while (i < recv[l].size() && recv[l][i].first == q) { gp.ghost.push_back(recv[l][i].second); ++i; }
      while (j < send[l].size() && send[l][j].first == q) { gp.send_elem.push_back(send[l][j].second); ++j; }
      gp.recv_off.push_back(gp.ghost.size());
      gp.send_off.push_back(gp.send_elem.size());
      // (layer 1: a face joins the two tets, so both directions exist)
      if (l == 0 && (gp.recv_off[gp.recv_off.size() - 2] == gp.recv_off.back() || gp.send_off[gp.send_off.size() - 2] == gp.send_off.back()))
        return fail(std::string(who) + ": a rank receives from us but sends nothing (asymmetric esuel)");
    }
    if (l == 0) gp.nghost1 = gp.ghost.size();
  }
  return 0;
}
}  // namespace

struct qdg_chunk {
  size_t nielem = 0, nunk = 0, nnode = 0, nghost1 = 0;
  int depth = 1;
  std::vector<size_t> inpoel;     // [4*nunk] local node ids
  std::vector<size_t> elem_gid;   // [nunk]
  std::vector<size_t> node_gid;   // [nnode]
  std::vector<int32_t> nbr_rank;  // per plan entry: layer-1 entries (ranks ascending), then layer-2 entries
  std::vector<int32_t> nbr_layer;
  std::vector<size_t> send_off, send_elem, recv_off;
};

extern "C" int qdg_chunk_build(size_t nelem, size_t nnode, const size_t* inpoel, const int* esuel,
                               const int32_t* part, int nparts, int rank, qdg_chunk** out)
{
  return qdg_chunk_build_depth(nelem, nnode, inpoel, esuel, part, nparts, rank, 1, out);
}

extern "C" int qdg_chunk_build_depth(size_t nelem, size_t nnode, const size_t* inpoel, const int* esuel,
                                     const int32_t* part, int nparts, int rank, int depth, qdg_chunk** out)
{
  QDG_TRY
  if (!inpoel || !part || !out) return fail("qdg_chunk_build: null argument");
  *out = nullptr;
  if (rank < 0 || rank >= nparts) return fail("qdg_chunk_build: rank outside [0, nparts)");
  if (depth != 1 && depth != 2) return fail("qdg_chunk_build: depth must be 1 or 2");
  std::vector<int> own_esuel;
  if (!esuel) {                       // face adjacency of the whole mesh (FaceData.cpp:19-41 -> genEsuelTet)
    own_esuel.resize(4 * nelem);
    if (int rc = qdg_gen_esuel(nelem, inpoel, own_esuel.data())) return rc;
    esuel = own_esuel.data();
  }
  std::unique_ptr<qdg_chunk> c(new qdg_chunk);
  c->depth = depth;
  // owned tets keep the input order (a serial run keeps the file's numbering too)
  std::vector<size_t> owned;
  for (size_t e = 0; e < nelem; ++e) {
    if (part[e] < 0 || part[e] >= nparts) return fail("qdg_chunk_build: part entry outside [0, nparts)");
    if (part[e] == rank) owned.push_back(e);
  }
  if (owned.empty()) return fail("qdg_chunk_build: this rank owns no element");
  // both sides order a pair's tets by global id: the sender's list for q IS the order in which q
  // stores the ghosts it gets from us (DG.cpp:1023-1031 sends m_ghostData[q] in the order q's
  // m_ghost map expects)
  GhostPlan gp;
  if (int rc = ghost_plan(nelem, esuel, part, nullptr, rank, depth, gp, "qdg_chunk_build")) return rc;
  c->nbr_rank = gp.entry_rank; c->nbr_layer = gp.entry_layer;
  c->recv_off = gp.recv_off; c->send_off = gp.send_off;
  c->nghost1 = gp.nghost1;
  c->nielem = owned.size();
  c->nunk = owned.size() + gp.ghost.size();
  c->elem_gid = owned;
  c->elem_gid.insert(c->elem_gid.end(), gp.ghost.begin(), gp.ghost.end());
  // global -> local ids
  std::vector<int> e_g2l(nelem, -1);
  for (size_t l = 0; l < c->nielem; ++l) e_g2l[owned[l]] = (int)l;
  c->send_elem.resize(gp.send_elem.size());
  for (size_t j = 0; j < gp.send_elem.size(); ++j) c->send_elem[j] = (size_t)e_g2l[gp.send_elem[j]];
  std::vector<int> n_g2l(nnode, -1);
  c->inpoel.resize(4 * c->nunk);
  for (size_t l = 0; l < c->nunk; ++l)
    for (int i = 0; i < 4; ++i) {
      const size_t g = inpoel[4 * c->elem_gid[l] + i];
      if (g >= nnode) return fail("qdg_chunk_build: inpoel entry out of range");
      if (n_g2l[g] < 0) { n_g2l[g] = (int)c->node_gid.size(); c->node_gid.push_back(g); }
      c->inpoel[4 * l + i] = (size_t)n_g2l[g];
    }
  c->nnode = c->node_gid.size();
  *out = c.release();
  return 0;
  QDG_CATCH
}

extern "C" int qdg_chunk_layers(const qdg_chunk* c, int32_t* depth, size_t* nghost1, int32_t* nbr_layer)
{
  QDG_TRY
  if (!c) return fail("qdg_chunk_layers: null chunk");
  if (depth) *depth = c->depth;
  if (nghost1) *nghost1 = c->nghost1;
  if (nbr_layer && !c->nbr_layer.empty()) std::memcpy(nbr_layer, c->nbr_layer.data(), c->nbr_layer.size() * sizeof(int32_t));
  return 0;
  QDG_CATCH
}

// the plan alone, for a caller that assembles its chunk itself (quinoa_amd/meshgen.py: the analytic block cut
// of the synthetic box hands in the tets around its block): tets are addressed by their index in the arrays
struct qdg_ghost_plan { GhostPlan gp; };

extern "C" int qdg_ghost_plan_build(size_t nelem, const int* esuel, const int32_t* owner, const size_t* gid, int rank,
                                    int depth, qdg_ghost_plan** out)
{
  QDG_TRY
  if (!esuel || !owner || !out) return fail("qdg_ghost_plan_build: null argument");
  *out = nullptr;
  if (depth != 1 && depth != 2) return fail("qdg_ghost_plan_build: depth must be 1 or 2");
  if (nelem > (size_t)INT32_MAX) return fail("qdg_ghost_plan_build: too many elements");
  std::unique_ptr<qdg_ghost_plan> p(new qdg_ghost_plan);
  if (int rc = ghost_plan(nelem, esuel, owner, gid, rank, depth, p->gp, "qdg_ghost_plan_build")) return rc;
  *out = p.release();
  return 0;
  QDG_CATCH
}

extern "C" int qdg_ghost_plan_sizes(const qdg_ghost_plan* p, size_t* nghost, size_t* nghost1, size_t* nentry, size_t* nsend)
{
  QDG_TRY
  if (!p || !nghost || !nghost1 || !nentry || !nsend) return fail("qdg_ghost_plan_sizes: null argument");
  *nghost = p->gp.ghost.size(); *nghost1 = p->gp.nghost1; *nentry = p->gp.entry_rank.size(); *nsend = p->gp.send_elem.size();
  return 0;
  QDG_CATCH
}

extern "C" int qdg_ghost_plan_get(const qdg_ghost_plan* p, size_t* ghost, int32_t* entry_rank, int32_t* entry_layer,
                                  size_t* recv_off, size_t* send_off, size_t* send_elem)
{
  QDG_TRY
  if (!p) return fail("qdg_ghost_plan_get: null plan");
  auto cp = [](auto* dst, const auto& v) { if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
  cp(ghost, p->gp.ghost); cp(entry_rank, p->gp.entry_rank); cp(entry_layer, p->gp.entry_layer);
  cp(recv_off, p->gp.recv_off); cp(send_off, p->gp.send_off); cp(send_elem, p->gp.send_elem);
  return 0;
  QDG_CATCH
}

extern "C" int qdg_ghost_plan_destroy(qdg_ghost_plan* p)
{
  QDG_TRY
  delete p;
  return 0;
  QDG_CATCH
}

extern "C" int qdg_chunk_sizes(const qdg_chunk* c, size_t* nielem, size_t* nunk, size_t* nnode, size_t* nnbr,
                               size_t* nsend)
{
  QDG_TRY
  if (!c || !nielem || !nunk || !nnode || !nnbr || !nsend) return fail("qdg_chunk_sizes: null argument");
  *nielem = c->nielem; *nunk = c->nunk; *nnode = c->nnode; *nnbr = c->nbr_rank.size();
  *nsend = c->send_elem.size();
  return 0;
  QDG_CATCH
}

extern "C" int qdg_chunk_get(const qdg_chunk* c, size_t* inpoel, size_t* elem_gid, size_t* node_gid,
                             int32_t* nbr_rank, size_t* send_off, size_t* send_elem, size_t* recv_off)
{
  QDG_TRY
  if (!c) return fail("qdg_chunk_get: null chunk");
  auto cp = [](auto* dst, const auto& v) { if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
  cp(inpoel, c->inpoel); cp(elem_gid, c->elem_gid); cp(node_gid, c->node_gid); cp(nbr_rank, c->nbr_rank);
  cp(send_off, c->send_off); cp(send_elem, c->send_elem); cp(recv_off, c->recv_off);
  return 0;
  QDG_CATCH
}

extern "C" int qdg_chunk_destroy(qdg_chunk* c)
{
  QDG_TRY
  delete c;
  return 0;
  QDG_CATCH
}

// ---------------------------------------------------------------------------------------
// Uniform 1:8 refinement of a tetrahedron mesh -- the only kind of mesh refinement the
// reference's DG scheme runs during time stepping (Refiner::dtref with amr::dtref_uniform,
// src/Inciter/Refiner.cpp:403-408; error-based refinement does nothing for element-centred
// schemes, :946-950).  Children and their local node order follow AMR::refinement_t::
// refine_one_to_eight (src/Inciter/AMR/refinement.hpp:425-536): for the tet (A,B,C,D) with
// edge midpoints AB..CD, four corner tets and four tets of the inner octahedron cut along its
// AC-BD diagonal, in an order that keeps the Jacobian positive.  A boundary triangle (a,b,c)
// becomes (a,ab,ac), (b,bc,ab), (c,ac,bc), (ab,bc,ac) on the same side set (Refiner::boundary
// regenerates the side sets of the children).  New nodes are the edge midpoints, numbered after
// the old ones in the order the tets meet their edges.
// run fn(begin, end, thread) over [0, n) on up to 16 host threads (contiguous ranges in order)
template <class F> static unsigned par_ranges(size_t n, F&& fn, size_t serial_below = 32768)
{
  unsigned nt = std::thread::hardware_concurrency();
  nt = std::max(1u, std::min(nt, 16u));
  if (n < serial_below) nt = 1;
  const size_t chunk = (n + nt - 1) / nt;
  if (nt == 1) { fn((size_t)0, n, 0u); return 1; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t) {
    const size_t b0 = std::min(n, t * chunk), e0 = std::min(n, b0 + chunk);
    th.emplace_back([&fn, b0, e0, t] { fn(b0, e0, t); });
  }
  for (auto& t : th) t.join();
  return nt;
}

extern "C" int qdg_refine_uniform(size_t nelem, size_t nnode, const size_t* inpoel, const double* x,
                                  const double* y, const double* z, size_t ntri, const size_t* tri,
                                  qdg_refined** out)
{
  QDG_TRY
  if (!inpoel || !x || !y || !z || !out || (ntri && !tri)) return fail("qdg_refine_uniform: null argument");
  *out = nullptr;
  if (nnode > (size_t)UINT32_MAX) return fail("qdg_refine_uniform: too many nodes");
  std::unique_ptr<qdg_refined> r(new qdg_refined);
  // edge (min,max) -> midpoint node: sort-based (no hashing: deterministic and cache friendly),
  // on all host cores: the edge slots are dealt into buckets by the edge's smaller node (a range
  // of node ids per bucket, so the buckets are in key order), each bucket sorted by one thread
  static const int EDG[6][2] = { {0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3} };   // AB AC AD BC BD CD
  struct E { uint64_t key; size_t slot; };
  const size_t ns = 6 * nelem;
  rawvec<E> ed(ns);
  const size_t NB = std::max<size_t>(1, std::min<size_t>(256, nnode / 64 + 1));
  auto bucket_of = [&](uint64_t key) { return (size_t)((key >> 32) * NB / std::max<size_t>(nnode, 1)); };
  std::vector<std::vector<size_t>> cnt;          // [thread][bucket]
  int bad = 0;
  std::vector<std::pair<size_t, size_t>> range;  // slot range of every thread
  {
    unsigned nt_used = 0;
    std::vector<std::vector<size_t>> c(16, std::vector<size_t>(NB, 0));
    std::vector<std::pair<size_t, size_t>> rg(16, { 0, 0 });
    std::vector<int> badt(16, 0);
    nt_used = par_ranges(nelem, [&](size_t e0, size_t e1, unsigned t) {
      rg[t] = { 6 * e0, 6 * e1 };
      for (size_t e = e0; e < e1; ++e)
        for (int k = 0; k < 6; ++k) {
          const size_t a = inpoel[4 * e + EDG[k][0]], b = inpoel[4 * e + EDG[k][1]];
          if (a >= nnode || b >= nnode) { badt[t] = 1; ed[6 * e + k] = { 0, 6 * e + k }; continue; }
          if (a == b) { badt[t] = 2; ed[6 * e + k] = { 0, 6 * e + k }; continue; }
          const uint64_t key = ((uint64_t)std::min(a, b) << 32) | (uint64_t)std::max(a, b);
          ed[6 * e + k] = { key, 6 * e + k };
          ++c[t][bucket_of(key)];
        }
    });
    for (unsigned t = 0; t < nt_used; ++t) bad = std::max(bad, badt[t]);
    c.resize(nt_used); rg.resize(nt_used);
    cnt.swap(c); range.swap(rg);
  }
  if (bad == 1) return fail("qdg_refine_uniform: inpoel entry out of range");
  if (bad == 2) return fail("qdg_refine_uniform: degenerate tet");
  // bucket offsets; within a bucket the threads' pieces follow each other in slot order
  std::vector<size_t> boff(NB + 1, 0);
  for (size_t b = 0; b < NB; ++b) { size_t n = 0; for (auto& c : cnt) n += c[b]; boff[b + 1] = boff[b] + n; }
  rawvec<E> sorted(ns);
  {
    std::vector<std::vector<size_t>> pos(cnt.size(), std::vector<size_t>(NB));
    for (size_t b = 0; b < NB; ++b) { size_t o = boff[b]; for (size_t t = 0; t < cnt.size(); ++t) { pos[t][b] = o; o += cnt[t][b]; } }
    std::vector<std::thread> th;
    for (size_t t = 0; t < cnt.size(); ++t)
      th.emplace_back([&, t] {
        for (size_t i = range[t].first; i < range[t].second; ++i) sorted[pos[t][bucket_of(ed[i].key)]++] = ed[i];
      });
    for (auto& t : th) t.join();
  }
  rawvec<size_t> mid(ns), first(ns);
  par_ranges(NB, [&](size_t b0, size_t b1, unsigned) {
    for (size_t b = b0; b < b1; ++b) {
      std::sort(sorted.begin() + boff[b], sorted.begin() + boff[b + 1],
                [](const E& p, const E& q) { return p.key < q.key || (p.key == q.key && p.slot < q.slot); });
      // the first slot (in tet order) that meets an edge numbers its midpoint
      for (size_t i = boff[b]; i < boff[b + 1];) {
        size_t j = i;
        while (j < boff[b + 1] && sorted[j].key == sorted[i].key) { first[sorted[j].slot] = sorted[i].slot; ++j; }
        i = j;
      }
    }
  }, 2);
  // midpoints numbered in slot order: count per range, offsets, assign
  size_t nn = nnode;
  {
    std::vector<size_t> nnew(17, 0);
    std::vector<std::pair<size_t, size_t>> rg(16, { 0, 0 });
    const unsigned nt_used = par_ranges(ns, [&](size_t s0, size_t s1, unsigned t) {
      rg[t] = { s0, s1 };
      size_t n = 0;
      for (size_t q = s0; q < s1; ++q) n += first[q] == q;
      nnew[t + 1] = n;
    });
    for (unsigned t = 0; t < nt_used; ++t) nnew[t + 1] += nnew[t];
    nn = nnode + nnew[nt_used];
    r->x.resize(nn); r->y.resize(nn); r->z.resize(nn);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt_used; ++t)
      th.emplace_back([&, t] {
        size_t id = nnode + nnew[t];
        for (size_t q = rg[t].first; q < rg[t].second; ++q)
          if (first[q] == q) {
            const size_t a = (size_t)(ed[q].key >> 32), b = (size_t)(ed[q].key & 0xffffffffu);
            mid[q] = id;
            r->x[id] = 0.5 * (x[a] + x[b]); r->y[id] = 0.5 * (y[a] + y[b]); r->z[id] = 0.5 * (z[a] + z[b]);
            ++id;
          }
      });
    for (auto& t : th) t.join();
    std::memcpy(r->x.data(), x, nnode * sizeof(double));
    std::memcpy(r->y.data(), y, nnode * sizeof(double));
    std::memcpy(r->z.data(), z, nnode * sizeof(double));
  }
  r->nnode = nn;
  r->inpoel.resize(32 * nelem); r->parent.resize(8 * nelem);
  par_ranges(nelem, [&](size_t e0, size_t e1, unsigned) {
    for (size_t e = e0; e < e1; ++e) {
      const size_t A = inpoel[4 * e], B = inpoel[4 * e + 1], C = inpoel[4 * e + 2], D = inpoel[4 * e + 3];
      size_t M[6];
      for (int k = 0; k < 6; ++k) M[k] = mid[first[6 * e + k]];
      const size_t AB = M[0], AC = M[1], AD = M[2], BC = M[3], BD = M[4], CD = M[5];
      const size_t ch[8][4] = { { A, AB, AC, AD }, { B, BC, AB, BD }, { C, AC, BC, CD }, { D, AD, CD, BD },
                                { BC, CD, AC, BD }, { AB, BD, AC, AD }, { AB, BC, AC, BD }, { AC, BD, CD, AD } };
      for (int k = 0; k < 8; ++k) {
        for (int i = 0; i < 4; ++i) r->inpoel[4 * (8 * e + k) + i] = ch[k][i];
        r->parent[8 * e + k] = e;
      }
    }
  });
  // boundary triangles: midpoints looked up among the tets' edges (the buckets are in key order,
  // so `sorted` is sorted as a whole)
  if (ntri) {
    auto find = [&](size_t a, size_t b, size_t& m) {
      const uint64_t key = ((uint64_t)std::min(a, b) << 32) | (uint64_t)std::max(a, b);
      auto it = std::lower_bound(sorted.begin(), sorted.end(), key, [](const E& p, uint64_t k) { return p.key < k; });
      if (it == sorted.end() || it->key != key) return false;
      m = mid[first[it->slot]];
      return true;
    };
    r->tri.resize(12 * ntri);
    std::vector<int> badt(16, 0);
    par_ranges(ntri, [&](size_t t0, size_t t1, unsigned th) {
      for (size_t t = t0; t < t1; ++t) {
        const size_t a = tri[3 * t], b = tri[3 * t + 1], c = tri[3 * t + 2];
        size_t ab = 0, bc = 0, ac = 0;
        if (a >= nnode || b >= nnode || c >= nnode || !find(a, b, ab) || !find(b, c, bc) || !find(a, c, ac)) { badt[th] = 1; continue; }
        const size_t ct[4][3] = { { a, ab, ac }, { b, bc, ab }, { c, ac, bc }, { ab, bc, ac } };
        for (int k = 0; k < 4; ++k)
          for (int i = 0; i < 3; ++i) r->tri[3 * (4 * t + k) + i] = ct[k][i];
      }
    });
    for (int v : badt) if (v) return fail("qdg_refine_uniform: a side-set triangle is not a face of the mesh");
  }
  *out = r.release();
  return 0;
  QDG_CATCH
}

// ---------------------------------------------------------------- refinement of one rank's chunk
// Uniform 1:8 refinement of ONE RANK's chunk of a decomposition, done by the rank alone (what
// DG::resizePostAMR has after the Refiner ran on a chare, src/Inciter/DG.cpp:1536-1612): owned and
// ghost tets are refined with the same pattern (edge midpoints coincide across the chunk boundary,
// so the refined mesh stays conforming); the children of the owned tets are the new owned tets;
// the new ghost layer is the set of children of OLD ghosts that share a face with a new owned tet.
// The halo plan follows without communication: a child of my tet A touches a child of rank q's
// tet B only where A and B touched, i.e. only where both ranks already hold the other's tet -- so
// my send list to q and q's new ghosts from me are the same set, and both sides order it by the
// global id of the child (8 * parent's global id + child number).
// (struct qdg_chunk_refined: qdg_host.hpp -- qdg_mesh_refine_chunk fills one from the device)

namespace {
struct FK { uint32_t a, b, c, idx; };      // sorted node triple of a face, index of (tet, local face)
inline bool fk_less(const FK& p, const FK& q)
{ return p.a != q.a ? p.a < q.a : p.b != q.b ? p.b < q.b : p.c != q.c ? p.c < q.c : p.idx < q.idx; }
inline bool fk_same(const FK& p, const FK& q) { return p.a == q.a && p.b == q.b && p.c == q.c; }
inline FK fk_of(size_t n0, size_t n1, size_t n2, size_t idx)
{
  uint32_t a = (uint32_t)n0, b = (uint32_t)n1, c = (uint32_t)n2;
  if (a > b) std::swap(a, b);
  if (b > c) std::swap(b, c);
  if (a > b) std::swap(a, b);
  return { a, b, c, (uint32_t)idx };
}
// sort by key on all host cores: buckets by ranges of the smallest node, one thread per bucket
template <class V> void fk_sort(V& v, size_t nnode)
{
  const size_t n = v.size();
  if (n < 65536) { std::sort(v.begin(), v.end(), fk_less); return; }
  const size_t NB = 256;
  std::vector<size_t> cnt(NB + 1, 0);
  auto bk = [&](const FK& f) { return (size_t)((uint64_t)f.a * NB / std::max<size_t>(nnode, 1)); };
  for (const FK& f : v) ++cnt[bk(f) + 1];
  for (size_t b = 0; b < NB; ++b) cnt[b + 1] += cnt[b];
  V w(n);
  { std::vector<size_t> pos(cnt.begin(), cnt.end() - 1); for (const FK& f : v) w[pos[bk(f)]++] = f; }
  par_ranges(NB, [&](size_t b0, size_t b1, unsigned) {
    for (size_t b = b0; b < b1; ++b) std::sort(w.begin() + cnt[b], w.begin() + cnt[b + 1], fk_less);
  }, 2);
  v.swap(w);
}
const int FACE_OF[4][3] = { { 1, 2, 3 }, { 2, 0, 3 }, { 3, 0, 1 }, { 0, 2, 1 } };
}  // namespace

extern "C" int qdg_refine_chunk(size_t nielem, size_t nunk, size_t nnode, const size_t* inpoel,
                                const double* x, const double* y, const double* z, const size_t* gid,
                                size_t ntri, const size_t* tri, const int32_t* tri_set, size_t nnbr,
                                const int32_t* nbr_rank, const size_t* recv_counts, qdg_chunk_refined** out)
{
  QDG_TRY
  if (!inpoel || !x || !y || !z || !gid || !out || (ntri && (!tri || !tri_set)) || (nnbr && (!nbr_rank || !recv_counts)))
    return fail("qdg_refine_chunk: null argument");
  *out = nullptr;
  if (nielem == 0 || nielem > nunk) return fail("qdg_refine_chunk: need 0 < nielem <= nunk");
  size_t nghost = 0;
  for (size_t i = 0; i < nnbr; ++i) {
    nghost += recv_counts[i];
    if (i > 0 && nbr_rank[i] <= nbr_rank[i - 1]) return fail("qdg_refine_chunk: neighbour ranks must be ascending");
  }
  if (nghost != nunk - nielem) return fail("qdg_refine_chunk: receive counts do not add up to the ghost count");
  for (size_t i = 0; i < 4 * nunk; ++i)
    if (inpoel[i] >= nnode) return fail("qdg_refine_chunk: inpoel entry out of range");

  // side-set triangles that are faces of an OWNED tet (a triangle can have its three nodes in the
  // chunk without being one; a ghost's boundary faces belong to its owner).  All eight children of
  // an owned tet are owned, so every child triangle below is a face of a kept, owned tet.
  std::vector<size_t> tri_in; std::vector<int32_t> set_in;
  if (ntri) {
    rawvec<FK> all(4 * nielem);
    par_ranges(nielem, [&](size_t e0, size_t e1, unsigned) {
      for (size_t e = e0; e < e1; ++e)
        for (int f = 0; f < 4; ++f)
          all[4 * e + f] = fk_of(inpoel[4 * e + FACE_OF[f][0]], inpoel[4 * e + FACE_OF[f][1]], inpoel[4 * e + FACE_OF[f][2]], 4 * e + f);
    });
    fk_sort(all, nnode);
    for (size_t t = 0; t < ntri; ++t) {
      if (tri[3 * t] >= nnode || tri[3 * t + 1] >= nnode || tri[3 * t + 2] >= nnode) continue;
      FK k = fk_of(tri[3 * t], tri[3 * t + 1], tri[3 * t + 2], 0);
      auto it = std::lower_bound(all.begin(), all.end(), k, fk_less);
      if (it != all.end() && fk_same(*it, k)) {
        tri_in.insert(tri_in.end(), tri + 3 * t, tri + 3 * t + 3);
        set_in.push_back(tri_set[t]);
      }
    }
  }
  qdg_refined* rr = nullptr;
  if (int rc = qdg_refine_uniform(nunk, nnode, inpoel, x, y, z, set_in.size(), tri_in.data(), &rr)) return rc;
  std::unique_ptr<qdg_refined> r(rr);
  if (r->nnode > (size_t)UINT32_MAX || 32 * nunk > (size_t)UINT32_MAX) return fail("qdg_refine_chunk: chunk too large");
  const rawvec<size_t>& i2 = r->inpoel;
  const size_t nown = 8 * nielem, nall = 8 * nunk;

  // A new ghost is a child of an old ghost that shares a face with a child of an owned tet; such a
  // face lies on an old owned | ghost face.  So only the children of the tets on the old interface
  // take part in the matching (a surface-sized set, not all 32 * nunk child faces).
  std::vector<char> on_iface(nunk, 0);
  {
    rawvec<FK> all(4 * nunk);
    par_ranges(nunk, [&](size_t e0, size_t e1, unsigned) {
      for (size_t e = e0; e < e1; ++e)
        for (int f = 0; f < 4; ++f)
          all[4 * e + f] = fk_of(inpoel[4 * e + FACE_OF[f][0]], inpoel[4 * e + FACE_OF[f][1]], inpoel[4 * e + FACE_OF[f][2]], 4 * e + f);
    });
    fk_sort(all, nnode);
    for (size_t i = 0; i + 1 < all.size(); ++i)
      if (fk_same(all[i], all[i + 1])) {
        const size_t ea = all[i].idx >> 2, eb = all[i + 1].idx >> 2;
        if ((ea < nielem) != (eb < nielem)) on_iface[ea] = on_iface[eb] = 1;
      }
  }
  std::vector<FK> freef;                      // faces of the owned children next to the old interface
  for (size_t ep = 0; ep < nielem; ++ep)
    if (on_iface[ep])
      for (size_t e = 8 * ep; e < 8 * ep + 8; ++e)
        for (int f = 0; f < 4; ++f)
          freef.push_back(fk_of(i2[4 * e + FACE_OF[f][0]], i2[4 * e + FACE_OF[f][1]], i2[4 * e + FACE_OF[f][2]], 4 * e + f));
  fk_sort(freef, r->nnode);
  // owner (index into nbr_rank) of every old ghost
  std::vector<int32_t> owner_idx(nunk - nielem);
  { size_t o = 0; for (size_t i = 0; i < nnbr; ++i) for (size_t k = 0; k < recv_counts[i]; ++k) owner_idx[o++] = (int32_t)i; }
  // ghost children with a face among the free ones -> new ghosts; the owned child on the other
  // side of that face goes into the send list of the ghost's owner
  struct G { int32_t owner; size_t cgid; uint32_t child; };
  std::vector<G> ghosts;
  std::vector<std::vector<std::pair<size_t, uint32_t>>> sends(nnbr);   // (global id of the child, owned child)
  {
    std::vector<char> taken(nall - nown, 0);
    for (size_t e = nown; e < nall; ++e) {
      if (!on_iface[r->parent[e]]) continue;
      const int32_t ow = owner_idx[r->parent[e] - nielem];
      for (int f = 0; f < 4; ++f) {
        const FK k = fk_of(i2[4 * e + FACE_OF[f][0]], i2[4 * e + FACE_OF[f][1]], i2[4 * e + FACE_OF[f][2]], 0);
        auto it = std::lower_bound(freef.begin(), freef.end(), k, fk_less);
        if (it != freef.end() && fk_same(*it, k)) {
          if (!taken[e - nown]) {
            taken[e - nown] = 1;
            ghosts.push_back({ ow, 8 * gid[r->parent[e]] + (e & 7), (uint32_t)e });
          }
          const uint32_t oc = it->idx >> 2;
          sends[ow].push_back({ 8 * gid[r->parent[oc]] + (oc & 7), oc });
        }
      }
    }
  }
  std::sort(ghosts.begin(), ghosts.end(), [](const G& p, const G& q) { return p.owner != q.owner ? p.owner < q.owner : p.cgid < q.cgid; });
  std::unique_ptr<qdg_chunk_refined> c(new qdg_chunk_refined);
  c->nielem = nown; c->nunk = nown + ghosts.size();
  c->recv_counts.assign(nnbr, 0);
  for (const G& g : ghosts) ++c->recv_counts[g.owner];
  c->send_off.assign(nnbr + 1, 0);
  for (size_t q = 0; q < nnbr; ++q) {
    auto& v = sends[q];
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    c->send_off[q + 1] = c->send_off[q] + v.size();
    for (auto& pr : v) c->send_list.push_back(pr.second);
  }
  // kept tets: owned children in order, then the new ghosts; nodes renumbered in ascending order
  // of their ids in the refined chunk
  std::vector<uint32_t> keep(c->nunk);
  for (size_t e = 0; e < nown; ++e) keep[e] = (uint32_t)e;
  for (size_t i = 0; i < ghosts.size(); ++i) keep[nown + i] = ghosts[i].child;
  std::vector<int64_t> g2l(r->nnode, -1);
  for (uint32_t e : keep) for (int i = 0; i < 4; ++i) g2l[i2[4 * (size_t)e + i]] = 0;
  size_t nn = 0;
  for (size_t n = 0; n < r->nnode; ++n) if (g2l[n] == 0) g2l[n] = (int64_t)nn++;
  c->nnode = nn;
  c->x.resize(nn); c->y.resize(nn); c->z.resize(nn);
  for (size_t n = 0; n < r->nnode; ++n) if (g2l[n] >= 0) { c->x[g2l[n]] = r->x[n]; c->y[g2l[n]] = r->y[n]; c->z[g2l[n]] = r->z[n]; }
  c->inpoel.resize(4 * c->nunk); c->gid.resize(c->nunk); c->parent.resize(c->nunk);
  par_ranges(c->nunk, [&](size_t k0, size_t k1, unsigned) {
    for (size_t k = k0; k < k1; ++k) {
      const size_t e = keep[k];
      for (int i = 0; i < 4; ++i) c->inpoel[4 * k + i] = (size_t)g2l[i2[4 * e + i]];
      c->gid[k] = 8 * gid[r->parent[e]] + (e & 7);
      c->parent[k] = r->parent[e];
    }
  });
  for (size_t t = 0; t < r->tri.size() / 3; ++t) {
    const int64_t a = g2l[r->tri[3 * t]], b = g2l[r->tri[3 * t + 1]], d = g2l[r->tri[3 * t + 2]];
    if (a >= 0 && b >= 0 && d >= 0) {
      c->tri.push_back((size_t)a); c->tri.push_back((size_t)b); c->tri.push_back((size_t)d);
      c->tri_set.push_back(set_in[t / 4]);
    }
  }
  *out = c.release();
  return 0;
  QDG_CATCH
}

// The same for a chunk with TWO ghost layers (qdg_chunk_build_depth(depth = 2); entries = the plan's (rank,
// layer) entries in order, recv_counts per entry): all tets of the chunk are refined; the new layers and the new
// plan come from the rule of qdg_chunk_build_depth (ghost_plan) applied to the children of the tets within two
// faces of the old owned | ghost interface -- a child lies at least as many faces from a foreign child as its
// parent from a foreign tet, so nothing farther in can enter a layer or a send list; owner of a child = owner of
// its parent, global id 8 * gid(parent) + k.  No communication: every rank derives matching lists.
extern "C" int qdg_refine_chunk_depth(size_t nielem, size_t nunk, size_t nnode, const size_t* inpoel,
                                      const double* x, const double* y, const double* z, const size_t* gid,
                                      size_t ntri, const size_t* tri, const int32_t* tri_set, size_t nentry,
                                      const int32_t* entry_rank, const size_t* recv_counts, int depth,
                                      qdg_chunk_refined** out)
{
  QDG_TRY
  if (depth == 1) {
    const int rc = qdg_refine_chunk(nielem, nunk, nnode, inpoel, x, y, z, gid, ntri, tri, tri_set, nentry, entry_rank,
                                    recv_counts, out);
    if (rc == 0) {
      (*out)->nbr_rank.assign(entry_rank, entry_rank + nentry);
      (*out)->nbr_layer.assign(nentry, 1);
      (*out)->nghost1 = (*out)->nunk - (*out)->nielem;
    }
    return rc;
  }
  if (depth != 2) return fail("qdg_refine_chunk_depth: depth must be 1 or 2");
  if (!inpoel || !x || !y || !z || !gid || !out || (ntri && (!tri || !tri_set)) || (nentry && (!entry_rank || !recv_counts)))
    return fail("qdg_refine_chunk_depth: null argument");
  *out = nullptr;
  if (nielem == 0 || nielem > nunk) return fail("qdg_refine_chunk_depth: need 0 < nielem <= nunk");
  size_t nghost = 0;
  for (size_t i = 0; i < nentry; ++i) nghost += recv_counts[i];
  if (nghost != nunk - nielem) return fail("qdg_refine_chunk_depth: receive counts do not add up to the ghost count");
  for (size_t i = 0; i < 4 * nunk; ++i)
    if (inpoel[i] >= nnode) return fail("qdg_refine_chunk_depth: inpoel entry out of range");
  // side-set triangles that are faces of an owned tet (as qdg_refine_chunk)
  std::vector<size_t> tri_in; std::vector<int32_t> set_in;
  if (ntri) {
    rawvec<FK> all(4 * nielem);
    for (size_t e = 0; e < nielem; ++e)
      for (int f = 0; f < 4; ++f)
        all[4 * e + f] = fk_of(inpoel[4 * e + FACE_OF[f][0]], inpoel[4 * e + FACE_OF[f][1]], inpoel[4 * e + FACE_OF[f][2]], 4 * e + f);
    fk_sort(all, nnode);
    for (size_t t = 0; t < ntri; ++t) {
      if (tri[3 * t] >= nnode || tri[3 * t + 1] >= nnode || tri[3 * t + 2] >= nnode) continue;
      FK k = fk_of(tri[3 * t], tri[3 * t + 1], tri[3 * t + 2], 0);
      auto it = std::lower_bound(all.begin(), all.end(), k, fk_less);
      if (it != all.end() && fk_same(*it, k)) { tri_in.insert(tri_in.end(), tri + 3 * t, tri + 3 * t + 3); set_in.push_back(tri_set[t]); }
    }
  }
  qdg_refined* rr = nullptr;
  if (int rc = qdg_refine_uniform(nunk, nnode, inpoel, x, y, z, set_in.size(), tri_in.data(), &rr)) return rc;
  std::unique_ptr<qdg_refined> r(rr);
  if (32 * nunk > (size_t)INT32_MAX) return fail("qdg_refine_chunk_depth: chunk too large");
  const rawvec<size_t>& i2 = r->inpoel;
  const size_t nown = 8 * nielem;
  // owner of every old tet: this rank = a value no neighbour has
  const int32_t ME = INT32_MIN;
  std::vector<int32_t> powner(nunk, ME);
  { size_t o = nielem; for (size_t i = 0; i < nentry; ++i) for (size_t k = 0; k < recv_counts[i]; ++k) powner[o++] = entry_rank[i]; }
  for (size_t i = 0; i < nentry; ++i) if (entry_rank[i] == ME) return fail("qdg_refine_chunk_depth: bad neighbour rank");
  // the zone: all ghosts and the owned tets within two faces of one
  std::vector<int> pes(4 * nunk);
  if (int rc = qdg_gen_esuel(nunk, inpoel, pes.data())) return rc;
  std::vector<char> zone(nunk, 0);
  for (size_t e = nielem; e < nunk; ++e) zone[e] = 1;
  for (int hop = 0; hop < 2; ++hop) {
    std::vector<size_t> add;
    for (size_t e = 0; e < nielem; ++e) {
      if (zone[e]) continue;
      for (int f = 0; f < 4; ++f) { const int nb = pes[4 * e + f]; if (nb >= 0 && zone[nb]) { add.push_back(e); break; } }
    }
    for (size_t e : add) zone[e] = 1;
  }
  std::vector<uint32_t> sub;                       // children of the zone's tets
  for (size_t p = 0; p < nunk; ++p) if (zone[p]) for (int k = 0; k < 8; ++k) sub.push_back((uint32_t)(8 * p + k));
  std::vector<size_t> sinp(4 * sub.size()), sgid(sub.size());
  std::vector<int32_t> sown(sub.size());
  for (size_t i = 0; i < sub.size(); ++i) {
    const size_t e = sub[i];
    for (int q = 0; q < 4; ++q) sinp[4 * i + q] = i2[4 * e + q];
    sgid[i] = 8 * gid[r->parent[e]] + (e & 7);
    sown[i] = powner[r->parent[e]];
  }
  std::vector<int> ses(4 * sub.size());
  if (int rc = qdg_gen_esuel(sub.size(), sinp.data(), ses.data())) return rc;
  GhostPlan gp;
  if (int rc = ghost_plan(sub.size(), ses.data(), sown.data(), sgid.data(), ME, 2, gp, "qdg_refine_chunk_depth")) return rc;
  std::unique_ptr<qdg_chunk_refined> c(new qdg_chunk_refined);
  c->nielem = nown; c->nunk = nown + gp.ghost.size();
  c->nbr_rank = gp.entry_rank; c->nbr_layer = gp.entry_layer; c->nghost1 = gp.nghost1;
  const size_t ne2 = gp.entry_rank.size();
  c->recv_counts.resize(ne2);
  for (size_t i = 0; i < ne2; ++i) c->recv_counts[i] = gp.recv_off[i + 1] - gp.recv_off[i];
  c->send_off = gp.send_off;
  c->send_list.resize(gp.send_elem.size());
  for (size_t j = 0; j < gp.send_elem.size(); ++j) c->send_list[j] = sub[gp.send_elem[j]];       // owned child = its new local id
  std::vector<uint32_t> keep(c->nunk);
  for (size_t e = 0; e < nown; ++e) keep[e] = (uint32_t)e;
  for (size_t i = 0; i < gp.ghost.size(); ++i) keep[nown + i] = sub[gp.ghost[i]];
  std::vector<int64_t> g2l(r->nnode, -1);
  for (uint32_t e : keep) for (int i = 0; i < 4; ++i) g2l[i2[4 * (size_t)e + i]] = 0;
  size_t nn = 0;
  for (size_t n = 0; n < r->nnode; ++n) if (g2l[n] == 0) g2l[n] = (int64_t)nn++;
  c->nnode = nn;
  c->x.resize(nn); c->y.resize(nn); c->z.resize(nn);
  for (size_t n = 0; n < r->nnode; ++n) if (g2l[n] >= 0) { c->x[g2l[n]] = r->x[n]; c->y[g2l[n]] = r->y[n]; c->z[g2l[n]] = r->z[n]; }
  c->inpoel.resize(4 * c->nunk); c->gid.resize(c->nunk); c->parent.resize(c->nunk);
  for (size_t k = 0; k < c->nunk; ++k) {
    const size_t e = keep[k];
    for (int i = 0; i < 4; ++i) c->inpoel[4 * k + i] = (size_t)g2l[i2[4 * e + i]];
    c->gid[k] = 8 * gid[r->parent[e]] + (e & 7);
    c->parent[k] = r->parent[e];
  }
  for (size_t t = 0; t < r->tri.size() / 3; ++t) {
    const int64_t a = g2l[r->tri[3 * t]], b = g2l[r->tri[3 * t + 1]], d = g2l[r->tri[3 * t + 2]];
    if (a >= 0 && b >= 0 && d >= 0) {
      c->tri.push_back((size_t)a); c->tri.push_back((size_t)b); c->tri.push_back((size_t)d);
      c->tri_set.push_back(set_in[t / 4]);
    }
  }
  *out = c.release();
  return 0;
  QDG_CATCH
}

// the plan of a refined chunk: entries, layer-1 ghosts; nbr_rank / nbr_layer sized by *nentry (NULL: sizes only)
extern "C" int qdg_chunk_refined_plan(const qdg_chunk_refined* c, size_t* nentry, size_t* nghost1, int32_t* nbr_rank,
                                      int32_t* nbr_layer)
{
  QDG_TRY
  if (!c) return fail("qdg_chunk_refined_plan: null handle");
  if (nentry) *nentry = c->nbr_rank.size();
  if (nghost1) *nghost1 = c->nghost1;
  auto cp = [](auto* dst, const auto& v) { if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
  cp(nbr_rank, c->nbr_rank); cp(nbr_layer, c->nbr_layer);
  return 0;
  QDG_CATCH
}

extern "C" int qdg_chunk_refined_sizes(const qdg_chunk_refined* c, size_t* nielem, size_t* nunk, size_t* nnode,
                                       size_t* ntri, size_t* nsend)
{
  QDG_TRY
  if (!c) return fail("qdg_chunk_refined_sizes: null handle");
  if (nielem) *nielem = c->nielem;
  if (nunk) *nunk = c->nunk;
  if (nnode) *nnode = c->nnode;
  if (ntri) *ntri = c->tri_set.size();
  if (nsend) *nsend = c->send_list.size();
  return 0;
  QDG_CATCH
}

extern "C" int qdg_chunk_refined_get(const qdg_chunk_refined* c, size_t* inpoel, size_t* gid, size_t* parent,
                                     double* x, double* y, double* z, size_t* tri, int32_t* tri_set,
                                     size_t* send_off, size_t* send_list, size_t* recv_counts)
{
  QDG_TRY
  if (!c) return fail("qdg_chunk_refined_get: null handle");
  auto cp = [](auto* dst, const auto& v) { if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
  cp(inpoel, c->inpoel); cp(gid, c->gid); cp(parent, c->parent); cp(x, c->x); cp(y, c->y); cp(z, c->z);
  cp(tri, c->tri); cp(tri_set, c->tri_set); cp(send_off, c->send_off); cp(send_list, c->send_list);
  cp(recv_counts, c->recv_counts);
  return 0;
  QDG_CATCH
}

extern "C" int qdg_chunk_refined_destroy(qdg_chunk_refined* c)
{
  QDG_TRY
  delete c;
  return 0;
  QDG_CATCH
}

extern "C" int qdg_refined_get(const qdg_refined* r, size_t* nnode, size_t* inpoel, size_t* parent,
                               double* x, double* y, double* z, size_t* tri)
{
  QDG_TRY
  if (!r) return fail("qdg_refined_get: null handle");
  r->wait();                                   // (a copy from the device may still be in flight)
  if (r->pending && !r->pending->error.empty()) return fail("qdg_refined_get: " + r->pending->error);
  if (nnode) *nnode = r->nnode;
  auto cp = [](auto* dst, const auto& v) { if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
  cp(inpoel, r->inpoel); cp(parent, r->parent); cp(x, r->x); cp(y, r->y); cp(z, r->z); cp(tri, r->tri);
  return 0;
  QDG_CATCH
}

extern "C" int qdg_refined_sizes(const qdg_refined* r, size_t* nelem, size_t* nnode, size_t* ntri)
{
  QDG_TRY
  if (!r) return fail("qdg_refined_sizes: null handle");
  r->wait();
  if (nelem) *nelem = r->inpoel.size() / 4;
  if (nnode) *nnode = r->nnode;
  if (ntri) *ntri = r->tri.size() / 3;
  return 0;
  QDG_CATCH
}

extern "C" int qdg_refined_tri_sets(const qdg_refined* r, int32_t* tri_set)
{
  QDG_TRY
  if (!r || !tri_set) return fail("qdg_refined_tri_sets: null argument");
  r->wait();
  if (r->tri_set.size() != r->tri.size() / 3)
    return fail("qdg_refined_tri_sets: this handle carries no side-set ids (it was made from the caller's own triangle list)");
  std::memcpy(tri_set, r->tri_set.data(), r->tri_set.size() * sizeof(int32_t));
  return 0;
  QDG_CATCH
}

extern "C" int qdg_refined_destroy(qdg_refined* r)
{
  QDG_TRY
  if (r) r->wait();
  delete r;
  return 0;
  QDG_CATCH
}
