// qdg_tables.hpp -- host-side construction of the constant quadrature/basis
// tables the kernels read (qdg_device.hpp: Tables<NDOF>, QuadTet).
//
// Quadrature rules: src/PDE/Integrate/Quadrature.cpp:16-339 (same points,
// weights and ORDER as the reference).  Basis: src/PDE/Integrate/Basis.cpp.
#pragma once
#include <cstring>
#include "qdg_device.hpp"
#include "qdg_host.hpp"

namespace qdg {

inline void host_basis(int ndof, double xi, double eta, double zeta, double* B)
{
  B[0] = 1.0;
  if (ndof > 1) {
    B[1] = 2.0 * xi + eta + zeta - 1.0;
    B[2] = 3.0 * eta + zeta - 1.0;
    B[3] = 4.0 * zeta - 1.0;
  }
  if (ndof > 4) {
    B[4] = 6.0 * xi * xi + eta * eta + zeta * zeta + 6.0 * xi * eta + 6.0 * xi * zeta
         + 2.0 * eta * zeta - 6.0 * xi - 2.0 * eta - 2.0 * zeta + 1.0;
    B[5] = 5.0 * eta * eta + zeta * zeta + 10.0 * xi * eta + 2.0 * xi * zeta
         + 6.0 * eta * zeta - 2.0 * xi - 6.0 * eta - 2.0 * zeta + 1.0;
    B[6] = 6.0 * zeta * zeta + 12.0 * xi * zeta + 6.0 * eta * zeta - 2.0 * xi - eta
         - 7.0 * zeta + 1.0;
    B[7] = 10.0 * eta * eta + zeta * zeta + 8.0 * eta * zeta - 8.0 * eta - 2.0 * zeta + 1.0;
    B[8] = 6.0 * zeta * zeta + 18.0 * eta * zeta - 3.0 * eta - 7.0 * zeta + 1.0;
    B[9] = 15.0 * zeta * zeta - 10.0 * zeta + 1.0;
  }
}

// dB_k/dxi_j, src/PDE/Integrate/Basis.cpp:77-265 (the reference-space part)
inline void host_dbasis(int ndof, double xi, double eta, double zeta, double g[3][10])
{
  for (int j = 0; j < 3; ++j) for (int k = 0; k < 10; ++k) g[j][k] = 0.0;
  if (ndof > 1) {
    g[0][1] = 2.0; g[1][1] = 1.0; g[2][1] = 1.0;
    g[0][2] = 0.0; g[1][2] = 3.0; g[2][2] = 1.0;
    g[0][3] = 0.0; g[1][3] = 0.0; g[2][3] = 4.0;
  }
  if (ndof > 4) {
    g[0][4] = 12.0 * xi + 6.0 * eta + 6.0 * zeta - 6.0;
    g[1][4] = 6.0 * xi + 2.0 * eta + 2.0 * zeta - 2.0;
    g[2][4] = 6.0 * xi + 2.0 * eta + 2.0 * zeta - 2.0;
    g[0][5] = 10.0 * eta + 2.0 * zeta - 2.0;
    g[1][5] = 10.0 * xi + 10.0 * eta + 6.0 * zeta - 6.0;
    g[2][5] = 2.0 * xi + 6.0 * eta + 2.0 * zeta - 2.0;
    g[0][6] = 12.0 * zeta - 2.0;
    g[1][6] = 6.0 * zeta - 1.0;
    g[2][6] = 12.0 * xi + 6.0 * eta + 12.0 * zeta - 7.0;
    g[0][7] = 0.0;
    g[1][7] = 20.0 * eta + 8.0 * zeta - 8.0;
    g[2][7] = 8.0 * eta + 2.0 * zeta - 2.0;
    g[0][8] = 0.0;
    g[1][8] = 18.0 * zeta - 3.0;
    g[2][8] = 18.0 * eta + 12.0 * zeta - 7.0;
    g[0][9] = 0.0;
    g[1][9] = 0.0;
    g[2][9] = 30.0 * zeta - 10.0;
  }
}

// triangle rules, Quadrature.cpp:261-339
inline void host_quad_tri(int ng, double c[][2], double* w)
{
  switch (ng) {
    case 1: c[0][0] = 1.0 / 3.0; c[0][1] = 1.0 / 3.0; w[0] = 1.0; break;
    case 3:
      c[0][0] = 2.0 / 3.0; c[0][1] = 1.0 / 6.0; w[0] = 1.0 / 3.0;
      c[1][0] = 1.0 / 6.0; c[1][1] = 2.0 / 3.0; w[1] = 1.0 / 3.0;
      c[2][0] = 1.0 / 6.0; c[2][1] = 1.0 / 6.0; w[2] = 1.0 / 3.0;
      break;
    case 6: {
      const double c1 = 0.816847572980459, c2 = 0.091576213509771, c3 = 0.091576213509771;
      const double c4 = 0.108103018168070, c5 = 0.445948490915965, c6 = 0.445948490915965;
      const double w1 = 0.054975870996713638 * 2.0, w2 = 0.1116907969117165 * 2.0;
      c[0][0] = c1; c[0][1] = c2; w[0] = w1;
      c[1][0] = c2; c[1][1] = c3; w[1] = w1;
      c[2][0] = c3; c[2][1] = c1; w[2] = w1;
      c[3][0] = c4; c[3][1] = c5; w[3] = w2;
      c[4][0] = c5; c[4][1] = c6; w[4] = w2;
      c[5][0] = c6; c[5][1] = c4; w[5] = w2;
      break;
    }
    default: break;
  }
}

// tetrahedron rules, Quadrature.cpp:16-259
inline void host_quad_tet(int ng, double c[][3], double* w)
{
  auto set = [&](int i, double a, double b, double d, double ww) {
    c[i][0] = a; c[i][1] = b; c[i][2] = d; w[i] = ww;
  };
  switch (ng) {
    case 1: set(0, 0.25, 0.25, 0.25, 1.0); break;
    case 4: {
      const double a1 = 0.5854101966249685, a2 = 0.1381966011250105;
      set(0, a2, a2, a2, 0.25); set(1, a1, a2, a2, 0.25);
      set(2, a2, a1, a2, 0.25); set(3, a2, a2, a1, 0.25);
      break;
    }
    case 5: {
      const double s = 1.0 / 6.0, w9 = 9.0 / 20.0;
      set(0, 0.25, 0.25, 0.25, -12.0 / 15.0);
      set(1, s, s, s, w9); set(2, 0.5, s, s, w9); set(3, s, 0.5, s, w9); set(4, s, s, 0.5, w9);
      break;
    }
    case 11: {
      const double c1 = 0.3994035761667992, c2 = 0.1005964238332008;
      const double c3 = 343.0 / 7500.0, c4 = 56.0 / 375.0;
      const double p = 11.0 / 14.0, q = 1.0 / 14.0;
      set(0, 0.25, 0.25, 0.25, -148.0 / 1875.0);
      set(1, p, q, q, c3); set(2, q, p, q, c3); set(3, q, q, p, c3); set(4, q, q, q, c3);
      set(5, c1, c1, c2, c4); set(6, c1, c2, c1, c4); set(7, c1, c2, c2, c4);
      set(8, c2, c1, c1, c4); set(9, c2, c1, c2, c4); set(10, c2, c2, c1, c4);
      break;
    }
    case 14: {
      const double a = 0.0673422422100983, b = 0.3108859192633005, cc = 0.7217942490673264;
      const double d = 0.0927352503108912, e = 0.4544962958743506, f = 0.0455037041256494;
      const double p = 0.1126879257180162, q = 0.0734930431163619, r = 0.0425460207770812;
      set(0, a, b, b, p); set(1, b, a, b, p); set(2, b, b, a, p); set(3, b, b, b, p);
      set(4, cc, d, d, q); set(5, d, cc, d, q); set(6, d, d, cc, q); set(7, d, d, d, q);
      set(8, e, e, f, r); set(9, e, f, e, r); set(10, e, f, f, r);
      set(11, f, e, e, r); set(12, f, e, f, r); set(13, f, f, e, r);
      break;
    }
    default: break;
  }
}

template <int NDOF> inline void fill_tables(Tables<NDOF>& T)
{
  std::memset(&T, 0, sizeof(T));
  constexpr int NGF = Tables<NDOF>::NGF, NGV = Tables<NDOF>::NGV;
  double tc[6][2], tw[6];
  host_quad_tri(NGF, tc, tw);
  for (int g = 0; g < NGF; ++g) {
    T.fw[g] = tw[g];
    // eval_gp for a triangle (Basis.cpp:24-48): shp = (1-a-b, a, b)
    T.fs[g][0] = 1.0 - tc[g][0] - tc[g][1];
    T.fs[g][1] = tc[g][0];
    T.fs[g][2] = tc[g][1];
    for (int lf = 0; lf < 4; ++lf) {
      double wn[4] = { 0, 0, 0, 0 };
      for (int j = 0; j < 3; ++j) wn[LPOFA[lf][j]] = T.fs[g][j];
      double B[10];
      host_basis(NDOF, wn[1], wn[2], wn[3], B);
      for (int k = 0; k < NDOF; ++k) T.fB[lf][g][k] = B[k];
    }
  }
  double vc[14][3], vw[14];
  host_quad_tet(NGV, vc, vw);
  for (int g = 0; g < NGV; ++g) {
    T.vw[g] = vw[g];
    for (int d = 0; d < 3; ++d) T.vc[g][d] = vc[g][d];
    double B[10], dB[3][10];
    host_basis(NDOF, vc[g][0], vc[g][1], vc[g][2], B);
    host_dbasis(NDOF, vc[g][0], vc[g][1], vc[g][2], dB);
    for (int k = 0; k < NDOF; ++k) {
      T.vB[g][k] = B[k];
      for (int j = 0; j < 3; ++j) T.vdB[g][j][k] = dB[j][k];
    }
  }
}

inline void fill_quadtet(QuadTet& Q, int ng)
{
  std::memset(&Q, 0, sizeof(Q));
  Q.ng = ng;
  host_quad_tet(ng, Q.c, Q.w);
}

}  // namespace qdg
