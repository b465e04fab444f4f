// qdg_dgpde.hpp -- C++ host layer above the C ABI (include/qdg.h): a class that
// models Inciter's `DGPDE` concept for compressible flow with the reference's
// exact method names, argument lists and const-ness, so that it can be
// registered in the reference's DGFactory under the same key and nothing above
// the PDE layer changes.
//
//   reference                                      here
//   ---------------------------------------------  -----------------------------
//   inciter::dg::CompFlow<Physics,Problem>         qdg::dg::CompFlowHIP<Physics,Problem>
//     (src/PDE/CompFlow/DGCompFlow.hpp:49-706)
//   DGPDE::initialize/lhs/rhs/dt signatures        identical (src/PDE/DGPDE.hpp:80-114)
//   tk::Fields, inciter::FaceData, UnsMesh::Coords  template-free aliases below:
//     inside Quinoa define QDG_WITH_QUINOA and the reference's own types are
//     used; standalone (tests, bench) the minimal mirrors in this header are.
//   tk::Exception via Throw()                       qdg::Exception (std::runtime_error)
//   WENO_P1 / Superbee_P1 free functions            qdg::limit() (DG.cpp:1251-1260)
//
// Per-chare device state: `g_dgpde` is one const object per PE shared by all
// chares, so the mesh handle cannot live in the PDE object by value.  The
// adapter keeps a handle cache keyed by the identity of the chare's `inpoel`
// storage (stable for the life of a DG chare between AMR/LB events, SURVEY
// 8b); `DG` calls release(inpoel) from its destructor / before resizePostAMR.
#ifndef QDG_DGPDE_HPP
#define QDG_DGPDE_HPP

#include <array>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "qdg.h"

// -- shared by both configurations ---------------------------------------------------------------
namespace qdg {
class Exception : public std::runtime_error {
 public:
  explicit Exception(const std::string& m) : std::runtime_error(m) {}
};
inline void check(int rc)
{
  if (rc != 0) throw Exception(std::string("qdg: ") + qdg_last_error());
}
}  // namespace qdg

// -- the ONLY configuration-dependent part: where real, Fields, FaceData and Coords come from.
// Inside a Quinoa build (-DQDG_WITH_QUINOA) they are the reference's own types; otherwise the
// mirrors below (same members) stand in, so everything after this block -- InputDeck, the Problem
// traits, DeviceDG and the CompFlowHIP / TransportHIP adapters -- is the SAME code in both
// configurations: what tests/cpp drives is what a Quinoa build compiles.
#ifdef QDG_WITH_QUINOA
#include "Fields.hpp"
#include "FaceData.hpp"
#include "UnsMesh.hpp"
namespace qdg {
using real = tk::real;
using Fields = tk::Fields;
using FaceData = inciter::FaceData;
using Coords = tk::UnsMesh::Coords;
}
#else
namespace qdg {

using real = double;
using Coords = std::array<std::vector<real>, 3>;   // tk::UnsMesh::Coords

// tk::Data<UnkEqComp> (src/Base/Data.hpp:36-566): vec[unknown*nprop + offset + component]
class Fields {
 public:
  using ncomp_t = std::size_t;
  Fields() = default;
  Fields(ncomp_t nu, ncomp_t np) : m_vec(nu * np), m_nunk(nu), m_nprop(np) {}
  real& operator()(ncomp_t unknown, ncomp_t component, ncomp_t offset)
  { return m_vec[unknown * m_nprop + offset + component]; }
  const real& operator()(ncomp_t unknown, ncomp_t component, ncomp_t offset) const
  { return m_vec[unknown * m_nprop + offset + component]; }
  ncomp_t nunk() const noexcept { return m_nunk; }
  ncomp_t nprop() const noexcept { return m_nprop; }
  void fill(real v) { std::fill(m_vec.begin(), m_vec.end(), v); }
  void resize(ncomp_t nu, real v = 0.0) { m_vec.resize(nu * m_nprop, v); m_nunk = nu; }
  std::vector<real>& data() { return m_vec; }
  const std::vector<real>& data() const { return m_vec; }
 private:
  std::vector<real> m_vec;
  ncomp_t m_nunk = 0, m_nprop = 0;
};

// inciter::FaceData (src/Inciter/FaceData.hpp:41-106): same constructor
// arguments and accessors; built by libqdg's host mirrors.
class FaceData {
 public:
  FaceData() = default;
  FaceData(const std::vector<std::size_t>& inpoel,
           const std::map<int, std::vector<std::size_t>>& bface,
           const std::vector<std::size_t>& triinpoel)
    : m_bface(bface), m_triinpoel(triinpoel)
  {
    const std::size_t ne = inpoel.size() / 4;
    std::size_t nb = 0;
    for (const auto& s : m_bface) nb += s.second.size();
    m_esuel.resize(4 * ne);
    check(qdg_gen_esuel(ne, inpoel.data(), m_esuel.data()));
    m_nipfac = qdg_gen_nipfac(ne, nb, m_esuel.data());
    m_inpofa.resize(3 * m_nipfac);
    check(qdg_gen_inpofa(ne, nb, inpoel.data(), m_triinpoel.data(), m_esuel.data(), m_inpofa.data()));
    m_belem.resize(nb);
    check(qdg_gen_belem(ne, nb, inpoel.data(), m_inpofa.data(), m_belem.data()));
    m_esuf.resize(2 * m_nipfac);
    check(qdg_gen_esuf(ne, nb, m_belem.data(), m_esuel.data(), m_esuf.data()));
  }
  const std::map<int, std::vector<std::size_t>>& Bface() const { return m_bface; }
  const std::vector<std::size_t>& Triinpoel() const { return m_triinpoel; }
  std::size_t Nbfac() const { std::size_t n = 0; for (const auto& s : m_bface) n += s.second.size(); return n; }
  const std::vector<int>& Esuel() const { return m_esuel; }
  std::vector<int>& Esuel() { return m_esuel; }
  std::size_t Nipfac() const { return m_nipfac; }
  const std::vector<std::size_t>& Inpofa() const { return m_inpofa; }
  std::vector<std::size_t>& Inpofa() { return m_inpofa; }
  const std::vector<std::size_t>& Belem() const { return m_belem; }
  const std::vector<int>& Esuf() const { return m_esuf; }
  std::vector<int>& Esuf() { return m_esuf; }
 private:
  std::map<int, std::vector<std::size_t>> m_bface;
  std::vector<std::size_t> m_triinpoel;
  std::vector<int> m_esuel;
  std::size_t m_nipfac = 0;
  std::vector<std::size_t> m_inpofa;
  std::vector<std::size_t> m_belem;
  std::vector<int> m_esuf;
};

// tk::genGeoFaceTri / tk::genGeoElemTet (src/Mesh/DerivedData.cpp:1292-1491)
inline Fields genGeoFaceTri(std::size_t nipfac, const std::vector<std::size_t>& inpofa, const Coords& coord)
{
  Fields g(nipfac, 7);
  check(qdg_gen_geoface(nipfac, inpofa.data(), coord[0].data(), coord[1].data(), coord[2].data(),
                        g.data().data()));
  return g;
}
inline Fields genGeoElemTet(const std::vector<std::size_t>& inpoel, const Coords& coord)
{
  Fields g(inpoel.size() / 4, 4);
  check(qdg_gen_geoelem(inpoel.size() / 4, inpoel.data(), coord[0].data(), coord[1].data(),
                        coord[2].data(), g.data().data()));
  return g;
}

}  // namespace qdg
#endif  // QDG_WITH_QUINOA

namespace qdg {

// what dg::CompFlow's constructor reads from g_inputdeck
// (src/PDE/CompFlow/DGCompFlow.hpp:80-93; defaults InputDeck.hpp:191-238)
struct InputDeck {
  std::size_t ndof = 1, rdof = 1;            // discr::ndof, rdof
  int flux = QDG_FLUX_HLLC;                  // discr::flux
  int limiter = QDG_LIMITER_NONE;            // discr::limiter
  real cweight = 1.0;                        // discr::cweight
  real cfl = 0.0, dt = 0.0;                  // discr::cfl | dt
  real gamma = 1.4, pstiff = 0.0, cv = 717.5;
  real alpha = 0.0, beta = 0.0, p0 = 0.0;
  real betax = 0.0, betay = 0.0, betaz = 0.0, r0 = 0.0, ce = 0.0, kappa = 0.0;   // nl_energy_growth
  std::vector<std::string> bcdir, bcsym, bcextrapolate;   // side set id strings, as parsed
  std::vector<std::string> bcinlet, bcoutlet;              // transport only (DGTransport.hpp:163-168)
  bool pref = false;                         // pref::pref (scheme pdg)
  real tolref = 0.1;                         // pref::tolref
  // dg::Transport: component::transport (number of scalars of the system, DGTransport.hpp:84-85) and
  // the shear_diff parameters param::transport::u0 | lambda | diffusivity (ShearDiff.cpp:43-45)
  std::size_t ncomp = 1;
  std::vector<real> u0, lambda, diffusivity;
  int device = 0;
};

namespace dg {

struct Euler { };   // Physics policy (src/PDE/CompFlow/Physics/DGEuler.hpp): no-op for DG

// Problem policies: only type() is needed on the device path
struct UserDefined    { static int type() noexcept { return QDG_PROBLEM_USER_DEFINED; } };
struct SodShocktube   { static int type() noexcept { return QDG_PROBLEM_SOD_SHOCKTUBE; } };
struct SedovBlastwave { static int type() noexcept { return QDG_PROBLEM_SEDOV_BLASTWAVE; } };
struct VorticalFlow   { static int type() noexcept { return QDG_PROBLEM_VORTICAL_FLOW; } };
struct TaylorGreen    { static int type() noexcept { return QDG_PROBLEM_TAYLOR_GREEN; } };
struct RotatedSodShocktube { static int type() noexcept { return QDG_PROBLEM_ROTATED_SOD_SHOCKTUBE; } };
struct NLEnergyGrowth { static int type() noexcept { return QDG_PROBLEM_NL_ENERGY_GROWTH; } };
struct RayleighTaylor { static int type() noexcept { return QDG_PROBLEM_RAYLEIGH_TAYLOR; } };
// Transport (src/PDE/Transport/Physics/DGAdvection.hpp, Problem/SlotCyl.hpp)
struct Advection { };
struct SlotCyl        { static int type() noexcept { return QDG_PROBLEM_SLOT_CYL; } };
struct CylAdvect      { static int type() noexcept { return QDG_PROBLEM_CYL_ADVECT; } };
struct GaussHump      { static int type() noexcept { return QDG_PROBLEM_GAUSS_HUMP; } };
struct ShearDiff      { static int type() noexcept { return QDG_PROBLEM_SHEAR_DIFF; } };

namespace detail {

// The members of a DGPDE model (src/PDE/DGPDE.hpp:205-259) on top of the C ABI,
// shared by the CompFlow and Transport adapters below.
template <int PDE>
class DeviceDG {
 public:
  using ncomp_t = std::size_t;

 protected:
  DeviceDG(ncomp_t c, const InputDeck& deck, int problem) : m_system(c), m_deck(deck), m_state(new State)
  {
    if (c != 0) throw Exception("qdg: only equation system 0 is supported");
    for (const auto& s : deck.bcdir) { m_bcset.push_back(std::stoi(s)); m_bctype.push_back(QDG_BC_DIRICHLET); }
    if (PDE == QDG_PDE_COMPFLOW)
      for (const auto& s : deck.bcsym) { m_bcset.push_back(std::stoi(s)); m_bctype.push_back(QDG_BC_SYMMETRY); }
    for (const auto& s : deck.bcextrapolate) { m_bcset.push_back(std::stoi(s)); m_bctype.push_back(QDG_BC_EXTRAPOLATE); }
    if (PDE == QDG_PDE_TRANSPORT) {
      for (const auto& s : deck.bcinlet) { m_bcset.push_back(std::stoi(s)); m_bctype.push_back(QDG_BC_INLET); }
      for (const auto& s : deck.bcoutlet) { m_bcset.push_back(std::stoi(s)); m_bctype.push_back(QDG_BC_OUTLET); }
    }
    qdg_config cfg{};
    cfg.struct_size = (int32_t)sizeof(qdg_config);
    cfg.pde = PDE; cfg.pref = deck.pref ? 1 : 0; cfg.tolref = deck.tolref;
    cfg.device = deck.device;
    cfg.ndof = (int32_t)deck.ndof; cfg.rdof = (int32_t)deck.rdof;
    cfg.flux = deck.flux; cfg.limiter = deck.limiter; cfg.problem = problem;
    cfg.nbc = (int32_t)m_bcset.size();
    cfg.bc_sideset = m_bcset.data(); cfg.bc_type = m_bctype.data();
    cfg.gamma = deck.gamma; cfg.pstiff = deck.pstiff; cfg.cv = deck.cv; cfg.cweight = deck.cweight;
    cfg.alpha = deck.alpha; cfg.beta = deck.beta; cfg.p0 = deck.p0; cfg.cfl = deck.cfl; cfg.dt = deck.dt;
    cfg.betax = deck.betax; cfg.betay = deck.betay; cfg.betaz = deck.betaz;
    cfg.r0 = deck.r0; cfg.ce = deck.ce; cfg.kappa = deck.kappa;
    if (PDE == QDG_PDE_TRANSPORT) {
      cfg.ncomp = (int32_t)deck.ncomp;
      if (problem == QDG_PROBLEM_SHEAR_DIFF) {
        // TransportProblemShearDiff::errchk (ShearDiff.cpp:92-113)
        if (deck.u0.size() != deck.ncomp || deck.lambda.size() != 2 * deck.ncomp ||
            deck.diffusivity.size() != 3 * deck.ncomp)
          throw Exception("Wrong number of advection-diffusion PDE parameters u0 / lambda / diffusivity");
        cfg.tr_u0 = deck.u0.data(); cfg.tr_lambda = deck.lambda.data(); cfg.tr_diffusivity = deck.diffusivity.data();
      }
    }
    check(qdg_ctx_create(&cfg, &m_state->ctx));
  }

 public:
  //! DGPDE::initialize (src/PDE/DGPDE.hpp:80-86).  DG::setup calls it before any rhs/dt
  //! (src/Inciter/DG.cpp:995-999) and its arguments carry no FaceData, so it does not need
  //! the chare's mesh handle: the L2 projection of the initial condition is evaluated on
  //! the device from inpoel, coord and L alone (Initialize.cpp:29-112 divides by L).
  void initialize(const Fields& L, const std::vector<std::size_t>& inpoel, const Coords& coord,
                  Fields& unk, real t, const std::size_t nielem) const
  {
    check(qdg_initialize_from(m_state->ctx, nielem, coord[0].size(), inpoel.data(), coord[0].data(),
                              coord[1].data(), coord[2].data(), L.data().data(), t, unk.data().data()));
  }

  //! DGPDE::lhs (src/PDE/DGPDE.hpp:89-90).  The mass matrix depends on geoElem
  //! only; before the chare's mesh is registered it is evaluated on the host.
  void lhs(const Fields& geoElem, Fields& l) const
  {
    const std::size_t nd = m_deck.ndof, ncomp = PDE == QDG_PDE_TRANSPORT ? m_deck.ncomp : 5;
    for (std::size_t e = 0; e < geoElem.nunk(); ++e)
      for (std::size_t c = 0; c < ncomp; ++c) {
        const real vol = geoElem(e, 0, 0);
        l(e, c * nd, 0) = vol;
        if (nd > 1) { l(e, c*nd+1, 0) = vol / 10.0; l(e, c*nd+2, 0) = vol * 3.0 / 10.0; l(e, c*nd+3, 0) = vol * 3.0 / 5.0; }
        if (nd > 4) {
          l(e, c*nd+4, 0) = vol / 35.0; l(e, c*nd+5, 0) = vol / 21.0; l(e, c*nd+6, 0) = vol / 14.0;
          l(e, c*nd+7, 0) = vol / 7.0;  l(e, c*nd+8, 0) = vol * 3.0 / 14.0; l(e, c*nd+9, 0) = vol * 3.0 / 7.0;
        }
      }
  }

  //! Register (upload) the mesh of one DG chare; also done lazily by rhs()/dt().
  void attach(const Fields& geoFace, const Fields& geoElem, const FaceData& fd,
              const std::vector<std::size_t>& inpoel, const Coords& coord) const
  {
    std::lock_guard<std::mutex> lock(m_state->mtx);
    const std::uint64_t sig = signature(inpoel, coord, fd);
    {
      // same storage, same sizes AND same content signature: the chare's mesh is already on the
      // device.  A vector re-allocated at the same address (AMR, migration, a new chare reusing a
      // freed block) is recognised by its sizes or, at equal sizes, by the sampled checksum of
      // connectivity, coordinates and face-element pairs, and re-uploaded.
      auto it = m_state->meshes.find(inpoel.data());
      if (it != m_state->meshes.end()) {
        if (it->second.nunk == inpoel.size() / 4 && it->second.nnode == coord[0].size() &&
            it->second.nfac == fd.Esuf().size() / 2 && it->second.sig == sig) return;
        qdg_mesh_destroy(it->second.h);
        m_state->meshes.erase(it);
      }
    }
    std::vector<int32_t> ids; std::vector<std::size_t> off{0}, faces;
    for (const auto& s : fd.Bface()) {
      ids.push_back(s.first);
      faces.insert(faces.end(), s.second.begin(), s.second.end());
      off.push_back(faces.size());
    }
    if (faces.empty()) faces.push_back(0);
    if (ids.empty()) ids.push_back(0);
    qdg_bface bf{ fd.Bface().size(), ids.data(), off.data(), faces.data() };
    qdg_mesh* m = nullptr;
    const std::size_t nielem = fd.Esuel().size() / 4, nunk = inpoel.size() / 4;
    check(qdg_mesh_upload(m_state->ctx, nielem, nunk, coord[0].size(), inpoel.data(), coord[0].data(),
                          coord[1].data(), coord[2].data(), fd.Nbfac(), fd.Esuf().size() / 2,
                          fd.Esuf().data(), fd.Esuel().data(), fd.Inpofa().data(),
                          geoFace.data().data(), geoElem.data().data(), &bf, &m));
    m_state->meshes[inpoel.data()] = Cached{ m, nunk, coord[0].size(), fd.Esuf().size() / 2, sig };
  }

  //! Content signature of a chare's mesh: ALL of inpoel (integer data: four interleaved multiply-xor
  //! lanes, ~1 ms per million tets -- a different chunk of the same size or a locally swapped
  //! connectivity in the same storage is always told apart), and FNV-1a over the sizes and up to 4096
  //! evenly spaced entries each of the three coordinate arrays (bit patterns) and esuf.  Nodes moved
  //! between the sampled entries with the connectivity unchanged are the one change it can miss: call
  //! release() when a chare's mesh changes (DG::resizePostAMR, migration), as INTEGRATION.md asks.
  static std::uint64_t signature(const std::vector<std::size_t>& inpoel, const Coords& coord, const FaceData& fd)
  {
    std::uint64_t h = 1469598103934665603ull;
    auto mix = [&h](std::uint64_t v) { for (int b = 0; b < 8; ++b) { h ^= (v >> (8 * b)) & 0xffu; h *= 1099511628211ull; } };
    auto sample = [&mix](const auto& a) {
      const std::size_t n = a.size(), step = n > 4096 ? n / 4096 : 1;
      mix(n);
      for (std::size_t i = 0; i < n; i += step) {
        std::uint64_t v = 0;
        std::memcpy(&v, &a[i], sizeof(a[i]) < 8 ? sizeof(a[i]) : 8);
        mix(v);
      }
      if (n) { std::uint64_t v = 0; std::memcpy(&v, &a[n - 1], sizeof(a[n - 1]) < 8 ? sizeof(a[n - 1]) : 8); mix(v); }
    };
    {
      std::uint64_t l[4] = { 0x9e3779b97f4a7c15ull, 0xbf58476d1ce4e5b9ull, 0x94d049bb133111ebull, 0x2545f4914f6cdd1dull };
      const std::size_t n = inpoel.size(), n4 = n / 4 * 4;
      for (std::size_t i = 0; i < n4; i += 4)
        for (int j = 0; j < 4; ++j) l[j] = (l[j] ^ (std::uint64_t)inpoel[i + j]) * 0x100000001b3ull + 0x632be59bd9b4e019ull;
      for (std::size_t i = n4; i < n; ++i) l[0] = (l[0] ^ (std::uint64_t)inpoel[i]) * 0x100000001b3ull + 0x632be59bd9b4e019ull;
      mix(n); mix(l[0]); mix(l[1]); mix(l[2]); mix(l[3]);
    }
    sample(coord[0]); sample(coord[1]); sample(coord[2]); sample(fd.Esuf());
    return h;
  }

  //! Forget a chare's mesh (DG dtor, before resizePostAMR / migration)
  void release(const std::vector<std::size_t>& inpoel) const
  {
    std::lock_guard<std::mutex> lock(m_state->mtx);
    auto it = m_state->meshes.find(inpoel.data());
    if (it != m_state->meshes.end()) { qdg_mesh_destroy(it->second.h); m_state->meshes.erase(it); }
  }

  //! DGPDE::rhs (src/PDE/DGPDE.hpp:93-104, dg::CompFlow::rhs DGCompFlow.hpp:130-195)
  void rhs(real t, const Fields& geoFace, const Fields& geoElem, const FaceData& fd,
           const std::vector<std::size_t>& inpoel, const Coords& coord, const Fields& U,
           const std::vector<std::size_t>& ndofel, Fields& R) const
  {
    attach(geoFace, geoElem, fd, inpoel, coord);
    if (m_deck.pref) check(qdg_ndofel_set(handle(inpoel), ndofel.data()));   // p-adaptive DG
    check(qdg_rhs(handle(inpoel), t, U.data().data(), R.data().data()));
  }

  //! DGPDE::dt (src/PDE/DGPDE.hpp:107-114, dg::CompFlow::dt DGCompFlow.hpp:206-406)
  real dt(const Coords& coord, const std::vector<std::size_t>& inpoel, const FaceData& fd,
          const Fields& geoFace, const Fields& geoElem, const std::vector<std::size_t>& ndofel,
          const Fields& U) const
  {
    attach(geoFace, geoElem, fd, inpoel, coord);
    if (m_deck.pref) check(qdg_ndofel_set(handle(inpoel), ndofel.data()));
    real v = 0.0;
    check(qdg_dt(handle(inpoel), U.data().data(), &v));
    return v;
  }

  //! WENO_P1 / Superbee_P1 with the inputs DG::lim hands the reference's free functions
  //! (src/Inciter/DG.cpp:1251-1260; Limiter.cpp:29-44, 155-175): works before the chare's
  //! mesh is on the device (the first DG::lim of a run precedes the first rhs/dt)
  void limit(const std::vector<int>& esuel, const std::vector<std::size_t>& /*inpoel*/,
             const std::vector<std::size_t>& ndofel, const Coords& /*coord*/, Fields& U) const
  {
    check(qdg_limit_from(m_state->ctx, esuel.size() / 4, U.nunk(), esuel.data(),
                         ndofel.empty() ? nullptr : ndofel.data(), U.data().data()));
  }
  //! the same through an attached mesh handle
  void limit(const std::vector<std::size_t>& inpoel, Fields& U) const
  { check(qdg_limit(handle(inpoel), U.data().data())); }
  //! Superbee_P1 with the per-element ndof of p-adaptive DG (Limiter.cpp:155-180)
  void limit(const std::vector<std::size_t>& inpoel, const std::vector<std::size_t>& ndofel, Fields& U) const
  {
    if (m_deck.pref) check(qdg_ndofel_set(handle(inpoel), ndofel.data()));
    check(qdg_limit(handle(inpoel), U.data().data()));
  }

  //! DGPDE::analyticSolution (src/PDE/DGPDE.hpp:141-144): the Problem's solution at a point
  std::vector<real> analyticSolution(real xi, real yi, real zi, real t) const
  {
    std::vector<real> s(PDE == QDG_PDE_TRANSPORT ? m_deck.ncomp : 5);
    check(qdg_solution(m_state->ctx, 1, &xi, &yi, &zi, t, s.data()));
    return s;
  }

  //! DGPDE::fieldNames (src/PDE/DGPDE.hpp:120-121): the Problem's own list
  //! (Problem::fieldNames, e.g. TaylorGreen.cpp:108-133: numerical, analytical and err fields)
  std::vector<std::string> fieldNames() const
  {
    std::size_t n = 0;
    check(qdg_ctx_field_count(m_state->ctx, &n));
    if (m_deck.pref && PDE == QDG_PDE_COMPFLOW) --n;   // "ndof" is named by dg::Transport only (DGTransport.hpp:225-226)
    std::vector<std::string> names;
    for (std::size_t f = 0; f < n; ++f) names.emplace_back(qdg_ctx_field_name(m_state->ctx, f));
    return names;
  }

  //! DGPDE::names: labels of the integral variables in the diagnostics file
  std::vector<std::string> names() const
  {
    if (PDE == QDG_PDE_TRANSPORT) {            // depvar + component (DGTransport.hpp:282-291)
      std::vector<std::string> n;
      for (std::size_t c = 0; c < m_deck.ncomp; ++c) n.push_back("c" + std::to_string(c));
      return n;
    }
    return { "r", "ru", "rv", "rw", "re" };
  }

  //! DGPDE::fieldOutput (src/PDE/DGPDE.hpp:126-131): Problem::fieldOutput evaluated on the
  //! device from the cell means of U and the centroids/volumes of geoElem, every field of
  //! fieldNames() (DGCompFlow.hpp:447-462, DGTransport.hpp:248-279)
  std::vector<std::vector<real>> fieldOutput(real t, const Fields& geoElem, Fields& U) const
  {
    const std::size_t n = U.nunk();
    std::size_t nf = 0;
    check(qdg_ctx_field_count(m_state->ctx, &nf));
    if (m_deck.pref) --nf;                      // the ndof column is added by DG::writeFields
    std::vector<real> flat(nf * n);
    check(qdg_field_output_from(m_state->ctx, t, n, geoElem.data().data(), U.data().data(), flat.data()));
    std::vector<std::vector<real>> out(nf);
    for (std::size_t f = 0; f < nf; ++f) out[f].assign(flat.begin() + f * n, flat.begin() + (f + 1) * n);
    return out;
  }

  //! DGPDE::avgElemToNode (src/PDE/DGPDE.hpp:134-139; dg::CompFlow::avgElemToNode,
  //! DGCompFlow.hpp:465-552; dg::Transport returns no nodal fields, DGTransport.hpp:231-239)
  std::vector<std::vector<real>> avgElemToNode(const std::vector<std::size_t>& inpoel, const Coords& coord,
                                               const Fields& /*geoElem*/, const Fields& U) const
  {
    std::vector<std::vector<real>> out;
    if (PDE == QDG_PDE_TRANSPORT) return out;
    const std::size_t nn = coord[0].size();
    std::vector<real> flat(6 * nn);
    check(qdg_avg_elem_to_node(m_state->ctx, inpoel.size() / 4, nn, inpoel.data(), U.data().data(), flat.data()));
    out.resize(6);
    for (std::size_t f = 0; f < 6; ++f) out[f].assign(flat.begin() + f * nn, flat.begin() + (f + 1) * nn);
    return out;
  }

  //! Problem::side via the configured BC lists (DGCompFlow.hpp:430-434)
  void side(std::unordered_set<int>& conf) const { for (int s : m_bcset) conf.insert(s); }

  //! the device handle of a chare's mesh, for the resident fast path (qdg_step etc.)
  qdg_mesh* handle(const std::vector<std::size_t>& inpoel) const
  {
    std::lock_guard<std::mutex> lock(m_state->mtx);
    auto it = m_state->meshes.find(inpoel.data());
    if (it == m_state->meshes.end()) throw Exception("qdg: mesh not attached");
    return it->second.h;
  }

 private:
  struct Cached { qdg_mesh* h; std::size_t nunk, nnode, nfac; std::uint64_t sig; };
  struct State {
    qdg_ctx* ctx = nullptr;
    std::mutex mtx;
    std::unordered_map<const std::size_t*, Cached> meshes;
    ~State()
    {
      for (auto& m : meshes) qdg_mesh_destroy(m.second.h);
      if (ctx) qdg_ctx_destroy(ctx);
    }
  };
  ncomp_t m_system;
  InputDeck m_deck;
  std::vector<int32_t> m_bcset, m_bctype;
  std::shared_ptr<State> m_state;   // copies of the PDE object share the device state
};

}  // namespace detail

//! Drop-in for dg::CompFlow (src/PDE/CompFlow/DGCompFlow.hpp:62-702)
template <class Physics, class Problem>
class CompFlowHIP : public detail::DeviceDG<QDG_PDE_COMPFLOW> {
 public:
  //! \param[in] c Equation system index (as dg::CompFlow; only system 0 is on the device)
  //! \param[in] deck The values dg::CompFlow reads from g_inputdeck
  explicit CompFlowHIP(ncomp_t c, const InputDeck& deck)
    : detail::DeviceDG<QDG_PDE_COMPFLOW>(c, deck, Problem::type()) {}
};

//! Drop-in for dg::Transport with one scalar (src/PDE/Transport/DGTransport.hpp:50-360);
//! deck.flux must be QDG_FLUX_UPWIND and deck.dt > 0
template <class Physics, class Problem>
class TransportHIP : public detail::DeviceDG<QDG_PDE_TRANSPORT> {
 public:
  explicit TransportHIP(ncomp_t c, const InputDeck& deck)
    : detail::DeviceDG<QDG_PDE_TRANSPORT>(c, deck, Problem::type()) {}
};

}  // namespace dg
}  // namespace qdg

#endif  // QDG_DGPDE_HPP
