// test_dgpde_model.cpp -- the adapter as a model of Inciter's type-erased DGPDE.
//
// inciter::DGPDE (src/PDE/DGPDE.hpp:43-262) holds any class T that has the
// members of its `Concept` behind a `Model<T>` (runtime concept idiom).  This
// driver restates that wrapper with the same virtual interface (:159-203) and
// forwarding members (:205-259), constructs it from qdg::dg::CompFlowHIP --
// which only compiles if the adapter has EVERY member with a compatible
// signature, avgElemToNode included -- and then drives it in the order the DG
// chare does at start-up, with no call outside the DGPDE interface before them:
//   DG::setup  (src/Inciter/DG.cpp:978-1007):  lhs -> initialize
//   DG::lim    (:1229-1260):                   limiter on the initial state
//   DG::dt     (:1360-1430), DG::solve (:1432-1508):  dt -> rhs
//   DG::writeFields (:1165-1215):              fieldNames, fieldOutput, avgElemToNode
// The mesh reaches the device when rhs()/dt() first see its FaceData.
//
//   usage: test_dgpde_model mesh.bin out.bin problem ndof
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <vector>

#include "qdg_dgpde.hpp"

namespace {

using qdg::Coords;
using qdg::FaceData;
using qdg::Fields;
using qdg::real;

// the runtime-concept wrapper, restated (DGPDE.hpp:43-262)
class DGPDE {
 public:
  template <class T> explicit DGPDE(T x) : self(new Model<T>(std::move(x))) {}
  // late-binding constructor used by the factory (DGPDE.hpp:74-77, PDEFactory.hpp:52-78)
  template <class T, class... Args>
  explicit DGPDE(std::function<T(Args...)> x, Args&&... args)
    : self(new Model<T>(std::move(x(std::forward<Args>(args)...)))) {}
  DGPDE(const DGPDE& x) : self(x.self->copy()) {}
  DGPDE(DGPDE&&) noexcept = default;

  void initialize(const Fields& L, const std::vector<std::size_t>& inpoel, const Coords& coord, Fields& unk,
                  real t, const std::size_t nielem) const
  { self->initialize(L, inpoel, coord, unk, t, nielem); }
  void lhs(const Fields& geoElem, Fields& l) const { self->lhs(geoElem, l); }
  void rhs(real t, const Fields& geoFace, const Fields& geoElem, const FaceData& fd,
           const std::vector<std::size_t>& inpoel, const Coords& coord, const Fields& U,
           const std::vector<std::size_t>& ndofel, Fields& R) const
  { self->rhs(t, geoFace, geoElem, fd, inpoel, coord, U, ndofel, R); }
  real dt(const Coords& coord, const std::vector<std::size_t>& inpoel, const FaceData& fd, const Fields& geoFace,
          const Fields& geoElem, const std::vector<std::size_t>& ndofel, const Fields& U) const
  { return self->dt(coord, inpoel, fd, geoFace, geoElem, ndofel, U); }
  void side(std::unordered_set<int>& conf) const { self->side(conf); }
  std::vector<std::string> fieldNames() const { return self->fieldNames(); }
  std::vector<std::string> names() const { return self->names(); }
  std::vector<std::vector<real>> fieldOutput(real t, const Fields& geoElem, Fields& U) const
  { return self->fieldOutput(t, geoElem, U); }
  std::vector<std::vector<real>> avgElemToNode(const std::vector<std::size_t>& inpoel, const Coords& coord,
                                               const Fields& geoElem, const Fields& U) const
  { return self->avgElemToNode(inpoel, coord, geoElem, U); }
  std::vector<real> analyticSolution(real xi, real yi, real zi, real t) const
  { return self->analyticSolution(xi, yi, zi, t); }

 private:
  struct Concept {
    virtual ~Concept() = default;
    virtual Concept* copy() const = 0;
    virtual void initialize(const Fields&, const std::vector<std::size_t>&, const Coords&, Fields&, real,
                            const std::size_t) const = 0;
    virtual void lhs(const Fields&, Fields&) const = 0;
    virtual void rhs(real, const Fields&, const Fields&, const FaceData&, const std::vector<std::size_t>&,
                     const Coords&, const Fields&, const std::vector<std::size_t>&, Fields&) const = 0;
    virtual real dt(const Coords&, const std::vector<std::size_t>&, const FaceData&, const Fields&, const Fields&,
                    const std::vector<std::size_t>&, const Fields&) const = 0;
    virtual void side(std::unordered_set<int>&) const = 0;
    virtual std::vector<std::string> fieldNames() const = 0;
    virtual std::vector<std::string> names() const = 0;
    virtual std::vector<std::vector<real>> fieldOutput(real, const Fields&, Fields&) const = 0;
    virtual std::vector<std::vector<real>> avgElemToNode(const std::vector<std::size_t>&, const Coords&,
                                                         const Fields&, const Fields&) const = 0;
    virtual std::vector<real> analyticSolution(real, real, real, real) const = 0;
  };
  template <class T> struct Model : Concept {
    explicit Model(T x) : data(std::move(x)) {}
    Concept* copy() const override { return new Model(*this); }
    void initialize(const Fields& L, const std::vector<std::size_t>& inpoel, const Coords& coord, Fields& unk,
                    real t, const std::size_t nielem) const override
    { data.initialize(L, inpoel, coord, unk, t, nielem); }
    void lhs(const Fields& geoElem, Fields& l) const override { data.lhs(geoElem, l); }
    void rhs(real t, const Fields& geoFace, const Fields& geoElem, const FaceData& fd,
             const std::vector<std::size_t>& inpoel, const Coords& coord, const Fields& U,
             const std::vector<std::size_t>& ndofel, Fields& R) const override
    { data.rhs(t, geoFace, geoElem, fd, inpoel, coord, U, ndofel, R); }
    real dt(const Coords& coord, const std::vector<std::size_t>& inpoel, const FaceData& fd, const Fields& geoFace,
            const Fields& geoElem, const std::vector<std::size_t>& ndofel, const Fields& U) const override
    { return data.dt(coord, inpoel, fd, geoFace, geoElem, ndofel, U); }
    void side(std::unordered_set<int>& conf) const override { data.side(conf); }
    std::vector<std::string> fieldNames() const override { return data.fieldNames(); }
    std::vector<std::string> names() const override { return data.names(); }
    std::vector<std::vector<real>> fieldOutput(real t, const Fields& geoElem, Fields& U) const override
    { return data.fieldOutput(t, geoElem, U); }
    std::vector<std::vector<real>> avgElemToNode(const std::vector<std::size_t>& inpoel, const Coords& coord,
                                                 const Fields& geoElem, const Fields& U) const override
    { return data.avgElemToNode(inpoel, coord, geoElem, U); }
    std::vector<real> analyticSolution(real xi, real yi, real zi, real t) const override
    { return data.analyticSolution(xi, yi, zi, t); }
    T data;
  };
  std::unique_ptr<Concept> self;
};

template <class T> std::vector<T> rd(FILE* f, size_t n)
{
  std::vector<T> v(n);
  if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
  return v;
}
template <class T> void wr(FILE* f, const std::vector<T>& v)
{
  uint64_t n = v.size();
  fwrite(&n, 8, 1, f);
  fwrite(v.data(), sizeof(T), v.size(), f);
}

// the DGFactory entry a maintainer registers (src/PDE/PDEFactory.hpp:52-78): ncomp_t -> DGPDE
template <class Eq> DGPDE make(const qdg::InputDeck& deck)
{
  std::function<Eq(std::size_t)> ctor = [deck](std::size_t c) { return Eq(c, deck); };
  return DGPDE(ctor, std::size_t(0));
}

struct Mesh {
  size_t nnode, nelem;
  Coords coord;
  std::vector<std::size_t> inpoel;
  std::map<int, std::vector<std::size_t>> bface;
  std::vector<std::size_t> triinpoel;
};

template <class Eq> int run(const Mesh& M, const qdg::InputDeck& deck, const char* outpath)
{
  const size_t nelem = M.nelem, ndof = deck.ndof;
  const auto& inpoel = M.inpoel;
  const auto& coord = M.coord;
  // what DG::DG builds (src/Inciter/DG.cpp:46-103)
  FaceData fd(inpoel, M.bface, M.triinpoel);
  auto geoFace = qdg::genGeoFaceTri(fd.Nipfac(), fd.Inpofa(), coord);
  auto geoElem = qdg::genGeoElemTet(inpoel, coord);

  Eq typed(0, deck);          // the object the factory lambda returns ...
  DGPDE eq = make<Eq>(deck);  // ... and the factory path itself (own device context)
  DGPDE eq2(eq);              // copies share the device state (g_dgpde is copied per PE)

  const size_t nprop = 5 * ndof;
  Fields L(nelem, nprop), U(nelem, nprop), R(nelem, nprop);
  std::vector<std::size_t> ndofel(nelem, ndof);
  // --- DG::setup order, through the type-erased interface only ---------------------------
  eq.lhs(geoElem, L);
  eq.initialize(L, inpoel, coord, U, 0.0, nelem);
  // --- DG::lim on the initial state (the reference calls the free functions WENO_P1 /
  //     Superbee_P1 here, DG.cpp:1251-1260): no mesh handle on the device yet --------------
  Fields Ulim = U;
  typed.limit(fd.Esuel(), inpoel, ndofel, coord, Ulim);
  // --- DG::dt, DG::solve --------------------------------------------------------------------
  const double dt = eq2.dt(coord, inpoel, fd, geoFace, geoElem, ndofel, Ulim);
  eq.rhs(0.0, geoFace, geoElem, fd, inpoel, coord, Ulim, ndofel, R);
  // --- DG::writeFields ----------------------------------------------------------------------
  const auto fnames = eq.fieldNames();
  auto fout = eq.fieldOutput(0.25, geoElem, Ulim);
  const auto nodal = eq.avgElemToNode(inpoel, coord, geoElem, Ulim);
  const auto asol = eq.analyticSolution(0.25, 0.5, 0.5, 0.0);
  std::unordered_set<int> conf;
  eq.side(conf);
  if (fout.size() != fnames.size() || nodal.size() != 6 || nodal[0].size() != M.nnode || asol.size() != 5 ||
      eq.names().size() != 5 || conf.size() != 6)
    throw std::runtime_error("output-side DGPDE members: wrong shapes");

  FILE* o = fopen(outpath, "wb");
  wr(o, L.data()); wr(o, U.data()); wr(o, Ulim.data()); wr(o, R.data());
  wr(o, std::vector<double>{ dt, (double)fnames.size() });
  for (const auto& v : fout) wr(o, v);
  for (const auto& v : nodal) wr(o, v);
  fclose(o);
  FILE* nf = fopen((std::string(outpath) + ".names").c_str(), "w");
  for (const auto& n : fnames) fprintf(nf, "%s\n", n.c_str());
  fclose(nf);
  printf("model ok: ndof %zu, %zu tets, %zu fields, dt=%.6e\n", ndof, nelem, fnames.size(), dt);
  return 0;
}

}  // namespace

int main(int argc, char** argv)
{
  if (argc != 5) { fprintf(stderr, "usage: %s mesh.bin out.bin problem ndof\n", argv[0]); return 2; }
  try {
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("mesh"); return 2; }
    auto hdr = rd<uint64_t>(f, 3);                      // nnode, nelem, ntri
    Mesh M;
    M.nnode = hdr[0]; M.nelem = hdr[1];
    const size_t ntri = hdr[2];
    for (int d = 0; d < 3; ++d) M.coord[d] = rd<double>(f, M.nnode);
    { auto v = rd<uint64_t>(f, 4 * M.nelem); M.inpoel.assign(v.begin(), v.end()); }
    std::vector<std::size_t> tri;
    { auto v = rd<uint64_t>(f, 3 * ntri); tri.assign(v.begin(), v.end()); }
    auto triset = rd<int32_t>(f, ntri);
    fclose(f);
    const std::string problem = argv[3];

    // boundary faces as the mesh loader regenerates them (Partitioner.cpp:357-393)
    M.triinpoel.resize(3 * ntri);
    std::vector<int32_t> fset(ntri);
    size_t nb = 0;
    qdg::check(qdg_bnd_faces(M.nelem, M.inpoel.data(), ntri, tri.data(), triset.data(), &nb,
                             M.triinpoel.data(), fset.data()));
    M.triinpoel.resize(3 * nb);
    for (size_t i = 0; i < nb; ++i) M.bface[fset[i]].push_back(i);

    qdg::InputDeck deck;
    deck.ndof = deck.rdof = (size_t)atoi(argv[4]);
    deck.cfl = 0.3;
    if (problem == "taylor_green") {
      deck.gamma = 5.0 / 3.0; deck.limiter = QDG_LIMITER_WENOP1; deck.cweight = 10.0;
      deck.bcdir = { "1", "2", "3", "4", "5", "6" };
      return run<qdg::dg::CompFlowHIP<qdg::dg::Euler, qdg::dg::TaylorGreen>>(M, deck, argv[2]);
    }
    deck.limiter = QDG_LIMITER_SUPERBEEP1;
    deck.bcextrapolate = { "1", "2" };
    deck.bcsym = { "3", "4", "5", "6" };
    return run<qdg::dg::CompFlowHIP<qdg::dg::Euler, qdg::dg::SodShocktube>>(M, deck, argv[2]);
  } catch (const std::exception& e) {
    fprintf(stderr, "FAILED: %s\n", e.what());
    return 1;
  }
}
