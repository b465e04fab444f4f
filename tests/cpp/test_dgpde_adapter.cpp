// test_dgpde_adapter.cpp -- drives the C++ DGPDE-shaped adapter
// (include/qdg_dgpde.hpp) the way Inciter's DG chare drives g_dgpde:
// FaceData ctor, geometry, lhs, initialize, rhs, dt, limiter, then two
// resident time steps.  Reads a mesh written by tests/test_gpu_cpp_adapter.py
// and writes the results for it to compare with the oracle.
//
//   usage: test_dgpde_adapter mesh.bin out.bin
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "qdg_dgpde.hpp"

template <class T> static std::vector<T> rd(FILE* f, size_t n)
{
  std::vector<T> v(n);
  if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
  return v;
}
template <class T> static void wr(FILE* f, const std::vector<T>& v)
{
  uint64_t n = v.size();
  fwrite(&n, 8, 1, f);
  fwrite(v.data(), sizeof(T), v.size(), f);
}

int main(int argc, char** argv)
{
  if (argc != 3) { fprintf(stderr, "usage: %s mesh.bin out.bin\n", argv[0]); return 2; }
  try {
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("mesh"); return 2; }
    auto hdr = rd<uint64_t>(f, 3);                      // nnode, nelem, ntri
    const size_t nnode = hdr[0], nelem = hdr[1], ntri = hdr[2];
    qdg::Coords coord;
    for (int d = 0; d < 3; ++d) coord[d] = rd<double>(f, nnode);
    std::vector<std::size_t> inpoel;
    { auto v = rd<uint64_t>(f, 4 * nelem); inpoel.assign(v.begin(), v.end()); }
    std::vector<std::size_t> tri;
    { auto v = rd<uint64_t>(f, 3 * ntri); tri.assign(v.begin(), v.end()); }
    auto triset = rd<int32_t>(f, ntri);
    fclose(f);

    // boundary faces as the mesh loader regenerates them (Partitioner.cpp:357-393)
    std::vector<std::size_t> triinpoel(3 * ntri);
    std::vector<int32_t> fset(ntri);
    size_t nb = 0;
    qdg::check(qdg_bnd_faces(nelem, inpoel.data(), ntri, tri.data(), triset.data(), &nb,
                             triinpoel.data(), fset.data()));
    triinpoel.resize(3 * nb);
    std::map<int, std::vector<std::size_t>> bface;
    for (size_t i = 0; i < nb; ++i) bface[fset[i]].push_back(i);

    // what DG::DG builds (src/Inciter/DG.cpp:46-103)
    qdg::FaceData fd(inpoel, bface, triinpoel);
    auto geoFace = qdg::genGeoFaceTri(fd.Nipfac(), fd.Inpofa(), coord);
    auto geoElem = qdg::genGeoElemTet(inpoel, coord);

    qdg::InputDeck deck;
    deck.ndof = deck.rdof = 4;
    deck.limiter = QDG_LIMITER_SUPERBEEP1;
    deck.cfl = 0.3;
    deck.bcextrapolate = { "1", "2" };
    deck.bcsym = { "3", "4", "5", "6" };
    qdg::dg::CompFlowHIP<qdg::dg::Euler, qdg::dg::SodShocktube> eq(0, deck);

    const size_t nprop = 20;
    qdg::Fields L(nelem, nprop), U(nelem, nprop), R(nelem, nprop);
    std::vector<std::size_t> ndofel(nelem, 4);
    eq.lhs(geoElem, L);
    eq.initialize(L, inpoel, coord, U, 0.0, nelem);      // before the mesh is on the device (DG::setup)
    eq.rhs(0.0, geoFace, geoElem, fd, inpoel, coord, U, ndofel, R);
    const double dt = eq.dt(coord, inpoel, fd, geoFace, geoElem, ndofel, U);
    qdg::Fields Ulim = U;
    eq.limit(inpoel, Ulim);

    // resident fast path: two full time steps
    qdg_mesh* m = eq.handle(inpoel);
    qdg::check(qdg_state_upload(m, U.data().data()));
    double t = 0.0, dts[2];
    for (int s = 0; s < 2; ++s) { qdg::check(qdg_step(m, t, 1e300, &dts[s])); t += dts[s]; }
    qdg::Fields U2(nelem, nprop);
    qdg::check(qdg_state_download(m, U2.data().data()));

    // error behaviour: a bad call reports, never crashes
    bool threw = false;
    try { qdg::Fields bad(1, 1); eq.limit(std::vector<std::size_t>{0, 1, 2, 3}, bad); }
    catch (const qdg::Exception&) { threw = true; }

    // output-side DGPDE members: fieldNames / names / fieldOutput / analyticSolution
    const auto fnames = eq.fieldNames();
    const auto dnames = eq.names();
    const auto fout = eq.fieldOutput(0.0, geoElem, U2);
    const auto asol = eq.analyticSolution(0.25, 0.5, 0.5, 0.0);      // Sod, left state
    if (fnames.size() != 6 || fout.size() != 6 || dnames.size() != 5 || asol.size() != 5 ||
        fnames[5] != "pressure_numerical" || fout[0].size() != nelem)
      throw std::runtime_error("output-side DGPDE members: wrong shapes");

    // dg::Transport stand-in (BASELINE config 1 physics) on the same chare data
    qdg::InputDeck tdeck;
    tdeck.ndof = tdeck.rdof = 1;
    tdeck.flux = QDG_FLUX_UPWIND;
    tdeck.dt = 5.0e-4;
    tdeck.bcdir = { "1", "2" }; tdeck.bcextrapolate = { "3", "4" };
    tdeck.bcinlet = { "5" }; tdeck.bcoutlet = { "6" };
    qdg::dg::TransportHIP<qdg::dg::Advection, qdg::dg::SlotCyl> tq(0, tdeck);
    qdg::Fields Lt(nelem, 1), Ut(nelem, 1), Rt(nelem, 1), Ut2(nelem, 1);
    std::vector<std::size_t> ndofel1(nelem, 1);
    tq.lhs(geoElem, Lt);
    tq.initialize(Lt, inpoel, coord, Ut, 0.0, nelem);
    tq.rhs(0.0, geoFace, geoElem, fd, inpoel, coord, Ut, ndofel1, Rt);
    const double tdt = tq.dt(coord, inpoel, fd, geoFace, geoElem, ndofel1, Ut);
    qdg_mesh* tm = tq.handle(inpoel);
    qdg::check(qdg_state_upload(tm, Ut.data().data()));
    double tt = 0.0, tdts[2];
    for (int s = 0; s < 2; ++s) { qdg::check(qdg_step(tm, tt, 1e300, &tdts[s])); tt += tdts[s]; }
    qdg::check(qdg_state_download(tm, Ut2.data().data()));

    // dg::Transport with two scalars, ShearDiff (component::transport 2; param::transport u0, lambda,
    // diffusivity per scalar), DG-P1, started at t0 = 1
    qdg::InputDeck sdeck;
    sdeck.ndof = sdeck.rdof = 4;
    sdeck.flux = QDG_FLUX_UPWIND;
    sdeck.dt = 2.0e-3;
    sdeck.bcdir = { "1", "2", "3", "4", "5", "6" };
    sdeck.ncomp = 2;
    sdeck.u0 = { 0.7, -0.3 }; sdeck.lambda = { 0.4, 0.1, -0.2, 0.3 };
    sdeck.diffusivity = { 3.0, 2.0, 1.0, 1.5, 2.5, 0.8 };
    qdg::dg::TransportHIP<qdg::dg::Advection, qdg::dg::ShearDiff> sq(0, sdeck);
    qdg::Fields Ls(nelem, 8), Us(nelem, 8), Us2(nelem, 8);
    sq.lhs(geoElem, Ls);
    sq.initialize(Ls, inpoel, coord, Us, 1.0, nelem);
    std::vector<std::size_t> ndofel4(nelem, 4);
    (void)sq.dt(coord, inpoel, fd, geoFace, geoElem, ndofel4, Us);    // attaches the chare's mesh (as DG::dt does)
    qdg_mesh* sm = sq.handle(inpoel);
    qdg::check(qdg_state_upload(sm, Us.data().data()));
    double st = 1.0, sdt = 0.0;
    for (int s = 0; s < 2; ++s) { qdg::check(qdg_step(sm, st, 1e300, &sdt)); st += sdt; }
    qdg::check(qdg_state_download(sm, Us2.data().data()));
    const auto ssol = sq.analyticSolution(0.3, 0.2, 0.1, 1.5);
    if (ssol.size() != 2 || sq.fieldNames().size() != 6) throw std::runtime_error("two-scalar Transport: wrong shapes");
    bool sthrew = false;
    try {
      qdg::InputDeck bad = sdeck; bad.lambda.pop_back();
      qdg::dg::TransportHIP<qdg::dg::Advection, qdg::dg::ShearDiff> b(0, bad);
    } catch (const qdg::Exception&) { sthrew = true; }
    if (!sthrew) throw std::runtime_error("ShearDiff errchk did not throw");

    FILE* o = fopen(argv[2], "wb");
    wr(o, L.data()); wr(o, U.data()); wr(o, R.data()); wr(o, Ulim.data()); wr(o, U2.data());
    wr(o, std::vector<double>{ dt, dts[0], dts[1], threw ? 1.0 : 0.0 });
    wr(o, Lt.data()); wr(o, Ut.data()); wr(o, Rt.data()); wr(o, Ut2.data());
    wr(o, std::vector<double>{ tdt, tdts[0], tdts[1] });
    wr(o, fout[0]); wr(o, fout[5]); wr(o, asol);
    wr(o, Us.data()); wr(o, Us2.data()); wr(o, ssol);
    fclose(o);
    sq.release(inpoel);
    tq.release(inpoel);
    eq.release(inpoel);
    printf("adapter ok: %zu tets, %zu faces, %zu boundary faces, dt=%.6e\n", nelem, fd.Nipfac(), nb, dt);
  } catch (const std::exception& e) {
    fprintf(stderr, "FAILED: %s\n", e.what());
    return 1;
  }
  return 0;
}
