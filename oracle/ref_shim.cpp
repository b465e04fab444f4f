// ref_shim.cpp -- extern "C" entry points onto the REFERENCE's own functions.
//
// TEST INFRASTRUCTURE ONLY (see dg_oracle.c).  This file is ours; it merely
// forwards plain-pointer arguments to the reference's tk::Jacobian,
// tk::inverseJacobian (src/Base/Vector.cpp:133-197) and
// tk::GaussQuadratureTet/Tri (src/PDE/Integrate/Quadrature.cpp:16-339), which
// the Makefile's `ref` target compiles from /root/reference where they lie.
// It exists so tests can check the C restatement in dg_oracle.c against the
// real reference code for those functions.  Built into oracle/_ref/ only.
#include <array>
#include <vector>
#include <cstddef>
#include "Vector.hpp"
#include "Quadrature.hpp"

static std::array<double,3> a3(const double* p) { return {{p[0], p[1], p[2]}}; }

extern "C" {

double ref_jacobian(const double* a, const double* b, const double* c, const double* d)
{ return tk::Jacobian(a3(a), a3(b), a3(c), a3(d)); }

void ref_inverse_jacobian(const double* a, const double* b, const double* c,
                          const double* d, double* out9)
{
  auto ji = tk::inverseJacobian(a3(a), a3(b), a3(c), a3(d));
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out9[3*i+j] = ji[i][j];
}

void ref_quad_tet(int ng, double* cx, double* cy, double* cz, double* w)
{
  std::array<std::vector<double>,3> c; std::vector<double> ww(ng);
  for (auto& v : c) v.resize(ng);
  tk::GaussQuadratureTet(static_cast<std::size_t>(ng), c, ww);
  for (int i = 0; i < ng; ++i) { cx[i]=c[0][i]; cy[i]=c[1][i]; cz[i]=c[2][i]; w[i]=ww[i]; }
}

void ref_quad_tri(int ng, double* cx, double* cy, double* w)
{
  std::array<std::vector<double>,2> c; std::vector<double> ww(ng);
  for (auto& v : c) v.resize(ng);
  tk::GaussQuadratureTri(static_cast<std::size_t>(ng), c, ww);
  for (int i = 0; i < ng; ++i) { cx[i]=c[0][i]; cy[i]=c[1][i]; w[i]=ww[i]; }
}

int ref_ngvol(int ndof)  { return (int)tk::NGvol((std::size_t)ndof); }
int ref_ngfa(int ndof)   { return (int)tk::NGfa((std::size_t)ndof); }
int ref_ngdiag(int ndof) { return (int)tk::NGdiag((std::size_t)ndof); }
int ref_nginit(int ndof) { return (int)tk::NGinit((std::size_t)ndof); }

}
