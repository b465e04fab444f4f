// qdg_devfn.hpp -- device-side helpers shared by the gfx950 kernel translation units of the
// DG compressible-flow path (qdg_rhs_p1.hip, qdg_rhs_p2.hip, qdg_kernels.hip): row I/O, basis,
// EoS, Riemann fluxes, Problem policies, geometry.  Every translation unit that includes this
// header owns a private copy of the constant-memory tables (no relocatable device code);
// upload_tables() in qdg_kernels.hip fills all of them through upload_tables_here().
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "qdg_device.hpp"
#include "qdg_kernels.hpp"
#include "qdg_tables.hpp"

#ifndef QDG_RCP_NR
#define QDG_RCP_NR 1    // Newton steps after v_rcp_f64 (1 step: R agrees with the fp64-division CPU result to 1e-15)
#endif
#ifndef QDG_SQRT_NR
#define QDG_SQRT_NR 1   // Goldschmidt steps after v_rsq_f64 (plus one residual correction)
#endif

namespace qdg {

static __constant__ Tables<1> c_tab1;
static __constant__ Tables<4> c_tab4;
static __constant__ Tables<10> c_tab10;
static __constant__ QuadTet c_qinit[3];   // NGinit rule per order index
static __constant__ QuadTet c_qdiag[3];   // NGdiag rule per order index

template <int NDOF> __device__ __forceinline__ const Tables<NDOF>& tab();
template <> __device__ __forceinline__ const Tables<1>& tab<1>() { return c_tab1; }
template <> __device__ __forceinline__ const Tables<4>& tab<4>() { return c_tab4; }
template <> __device__ __forceinline__ const Tables<10>& tab<10>() { return c_tab10; }

template <int NDOF> constexpr int order_index() { return NDOF == 1 ? 0 : NDOF == 4 ? 1 : 2; }

// ------------------------------------------------------------------ basics

// Field layout in HBM: element-major rows, U[e*NPROP + c*NDOF + k] -- the same
// order as the reference's tk::Fields rows, in device element numbering.  A
// face-neighbour gather then touches the 160 (P1) contiguous bytes of one row
// (1.25 cache lines) instead of 20 different lines of 20 separate planes.
__device__ __forceinline__ size_t fidx(int p, int e, int nprop) { return (size_t)e * nprop + p; }

// whole row of element e into registers with 16-byte loads (rows are 16-byte
// aligned when NPROP is even: P1 160 B, P2 400 B; P0 rows are 40 B)
template <int NPROP>
__device__ __forceinline__ void load_row(const double* __restrict__ U, int e, double* r)
{
  const double* p = U + (size_t)e * NPROP;
  if constexpr (NPROP % 2 == 0) {
    const double2* q = reinterpret_cast<const double2*>(__builtin_assume_aligned(p, 16));
#pragma unroll
    for (int i = 0; i < NPROP / 2; ++i) { const double2 v = q[i]; r[2 * i] = v.x; r[2 * i + 1] = v.y; }
  } else {
#pragma unroll
    for (int i = 0; i < NPROP; ++i) r[i] = p[i];
  }
}
template <int NPROP>
__device__ __forceinline__ void store_row(double* __restrict__ U, int e, const double* r)
{
  double* p = U + (size_t)e * NPROP;
  if constexpr (NPROP % 2 == 0) {
    double2* q = reinterpret_cast<double2*>(__builtin_assume_aligned(p, 16));
#pragma unroll
    for (int i = 0; i < NPROP / 2; ++i) q[i] = make_double2(r[2 * i], r[2 * i + 1]);
  } else {
#pragma unroll
    for (int i = 0; i < NPROP; ++i) p[i] = r[i];
  }
}

// non-temporal 16-byte store (global_store_dwordx4 ... nt): rows that are written once per launch and read by a
// later kernel.  (Non-temporal LOADS of streamed rows were measured and lose: the neighbour gathers of the same
// launch live off those lines in the L2 -- profiles/r04_limiter_experiments.log.)
__device__ __forceinline__ void store_nt(double2* p, double2 v)
{
  __builtin_nontemporal_store(v.x, &p->x);
  __builtin_nontemporal_store(v.y, &p->y);
}

// ---- coalesced row I/O of a 256-tet tile through LDS -------------------------
// The rows of a workgroup's 256 consecutive tets are one contiguous span of
// 256*NPROP doubles.  These helpers move that span with unit-stride 16-byte
// accesses (full 1 KiB wave transactions) instead of a 160-byte (P1) lane
// stride, and keep it in LDS, where in-tile face neighbours (about three
// quarters of all neighbours of a Morton-ordered tile) can be read without
// touching L1/L2.  Rows beyond `nrows` get benign filler.  All 256 threads call.
template <int NPROP>
__device__ __forceinline__ void tile_stage_rows(const double* __restrict__ U, int tile_e0, int nrows,
                                                double* __restrict__ lds)
{
  static_assert(NPROP % 2 == 0, "row staging needs 16-byte rows");
  const int tid = threadIdx.x;
  const double2* src = reinterpret_cast<const double2*>(U + (size_t)tile_e0 * NPROP);
  double2* dst = reinterpret_cast<double2*>(lds);
  const int nvalid = (nrows - tile_e0 < 256 ? nrows - tile_e0 : 256) * (NPROP / 2);
#pragma unroll
  for (int j = 0; j < NPROP / 2; ++j) {
    const int i = j * 256 + tid;
    dst[i] = (i < nvalid) ? src[i] : make_double2(1.0, 1.0);
  }
  __syncthreads();
}

template <int NPROP>
__device__ __forceinline__ void lds_row(const double* __restrict__ lds, int r, double* out)
{
  const double2* q = reinterpret_cast<const double2*>(lds + (size_t)r * NPROP);
#pragma unroll
  for (int j = 0; j < NPROP / 2; ++j) { const double2 v = q[j]; out[2 * j] = v.x; out[2 * j + 1] = v.y; }
}

// rows of the tile back to HBM, coalesced (every lane first deposits its row)
template <int NPROP>
__device__ __forceinline__ void tile_store_rows(double* __restrict__ U, int tile_e0, int nrows,
                                                double* __restrict__ lds, const double* r)
{
  const int tid = threadIdx.x;
  __syncthreads();            // all readers of the staged rows are done
  double2* row = reinterpret_cast<double2*>(lds + (size_t)tid * NPROP);
#pragma unroll
  for (int j = 0; j < NPROP / 2; ++j) row[j] = make_double2(r[2 * j], r[2 * j + 1]);
  __syncthreads();
  const double2* src = reinterpret_cast<const double2*>(lds);
  double2* dst = reinterpret_cast<double2*>(U + (size_t)tile_e0 * NPROP);
  const int nvalid = (nrows - tile_e0 < 256 ? nrows - tile_e0 : 256) * (NPROP / 2);
#pragma unroll
  for (int j = 0; j < NPROP / 2; ++j) {
    const int i = j * 256 + tid;
    if (i < nvalid) dst[i] = src[i];
  }
}

// XCD-aware workgroup -> element-tile map.  Workgroups are dealt round-robin
// over the 8 XCDs (b and b+8 share an XCD and its private 4 MiB L2), while a
// tet's face neighbours sit close to it in the Morton-ordered numbering.
// Giving XCD x the contiguous tile range [x*n/8, (x+1)*n/8) keeps the
// neighbour gathers inside one L2 instead of re-fetching the same DOFs through
// the fabric once per XCD.  Bijective for any grid size; placement affects
// speed only, never results.
__device__ __forceinline__ int xcd_tile(int bid, int nwg)
{
  constexpr int NXCD = 8;
  const int per = nwg / NXCD, rem = nwg - per * NXCD;
  const int xcd = bid % NXCD, idx = bid / NXCD;
  return xcd * per + (xcd < rem ? xcd : rem) + idx;
}

// Dubiner basis, src/PDE/Integrate/Basis.cpp:267-307
template <int NDOF>
__device__ __forceinline__ void eval_basis(double xi, double eta, double zeta, double* B)
{
  B[0] = 1.0;
  if constexpr (NDOF > 1) {
    B[1] = 2.0 * xi + eta + zeta - 1.0;
    B[2] = 3.0 * eta + zeta - 1.0;
    B[3] = 4.0 * zeta - 1.0;
  }
  if constexpr (NDOF > 4) {
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

// src/PDE/EoS/EoS.hpp:66-84
__device__ __forceinline__ double eos_pressure(const Phys& ph, double rho, double u, double v,
                                               double w, double rhoE)
{
  return (rhoE - 0.5 * rho * (u * u + v * v + w * w) - ph.pstiff) * (ph.gamma - 1.0) - ph.pstiff;
}
// src/PDE/EoS/EoS.hpp:95-108
__device__ __forceinline__ double eos_soundspeed(const Phys& ph, double rho, double pr)
{
  return sqrt(ph.gamma * (pr + ph.pstiff) / rho);
}
// src/PDE/EoS/EoS.hpp:123-140
__device__ __forceinline__ double eos_totalenergy(const Phys& ph, double rho, double u,
                                                  double v, double w, double pr)
{
  return (pr + ph.pstiff) / (ph.gamma - 1.0) + 0.5 * rho * (u * u + v * v + w * w) + ph.pstiff;
}

// HLLC, src/PDE/Integrate/Riemann/HLLC.hpp:36-125.  The 4-way branch is
// evaluated as per-lane selects (no wave divergence).
__device__ __forceinline__ void flux_hllc(const Phys& ph, const double* fn, const double* L,
                                          const double* R, double* flx)
{
  const double rhol = L[0], rhor = R[0];
  const double irl = 1.0 / rhol, irr = 1.0 / rhor;
  const double ul = L[1] * irl, vl = L[2] * irl, wl = L[3] * irl;
  const double ur = R[1] * irr, vr = R[2] * irr, wr = R[3] * irr;
  const double pl = eos_pressure(ph, rhol, ul, vl, wl, L[4]);
  const double pr = eos_pressure(ph, rhor, ur, vr, wr, R[4]);
  const double al = eos_soundspeed(ph, rhol, pl);
  const double ar = eos_soundspeed(ph, rhor, pr);
  const double vnl = ul * fn[0] + vl * fn[1] + wl * fn[2];
  const double vnr = ur * fn[0] + vr * fn[1] + wr * fn[2];
  const double rlr = sqrt(rhor * irl);
  const double irlr1 = 1.0 / (1.0 + rlr);
  const double vnroe = (vnr * rlr + vnl) * irlr1;
  const double aroe = (ar * rlr + al) * irlr1;
  const double Sl = fmin(vnl - al, vnroe - aroe);
  const double Sr = fmax(vnr + ar, vnroe + aroe);
  const double Sm = (rhor * vnr * (Sr - vnr) - rhol * vnl * (Sl - vnl) + pl - pr)
                  / (rhor * (Sr - vnr) - rhol * (Sl - vnl));
  const double pStar = rhol * (vnl - Sl) * (vnl - Sm) + pl;
  // branch ladder of HLLC.hpp:93-124 as per-lane predicates:
  //   Sl > 0 -> left flux; else Sm > 0 -> left star; else Sr >= 0 -> right star;
  //   else right flux
  // (every comparison of the reference is kept: with a NaN wave speed -- e.g. a
  // negative pressure at a Gauss point next to a strong shock -- all of them
  // are false and the reference falls through to the right-state flux)
  const bool c1 = Sl > 0.0;
  const bool c2 = !c1 && (Sl <= 0.0) && (Sm > 0.0);
  const bool c3 = !c1 && !c2 && (Sm <= 0.0) && (Sr >= 0.0);
  const bool left = c1 || c2;
  const bool star = c2 || c3;
  const double S = left ? Sl : Sr;
  const double vn = left ? vnl : vnr;
  const double p = left ? pl : pr;
  const double u0 = left ? L[0] : R[0], u1 = left ? L[1] : R[1], u2 = left ? L[2] : R[2],
               u3 = left ? L[3] : R[3], u4 = left ? L[4] : R[4];
  if (star) {
    const double id = 1.0 / (S - Sm);
    const double sv = S - vn, dp = pStar - p;
    const double s0 = sv * u0 * id;
    const double s1 = (sv * u1 + dp * fn[0]) * id;
    const double s2 = (sv * u2 + dp * fn[1]) * id;
    const double s3 = (sv * u3 + dp * fn[2]) * id;
    const double s4 = (sv * u4 - p * vn + pStar * Sm) * id;
    flx[0] = s0 * Sm;
    flx[1] = s1 * Sm + pStar * fn[0];
    flx[2] = s2 * Sm + pStar * fn[1];
    flx[3] = s3 * Sm + pStar * fn[2];
    flx[4] = (s4 + pStar) * Sm;
  } else {
    flx[0] = u0 * vn;
    flx[1] = u1 * vn + p * fn[0];
    flx[2] = u2 * vn + p * fn[1];
    flx[3] = u3 * vn + p * fn[2];
    flx[4] = (u4 + p) * vn;
  }
}

// Lax-Friedrichs, src/PDE/Integrate/Riemann/LaxFriedrichs.hpp:34-88
__device__ __forceinline__ void flux_lf(const Phys& ph, const double* fn, const double* L,
                                        const double* R, double* flx)
{
  const double rhol = L[0], rhor = R[0];
  const double ul = L[1] / rhol, vl = L[2] / rhol, wl = L[3] / rhol;
  const double ur = R[1] / rhor, vr = R[2] / rhor, wr = R[3] / rhor;
  const double pl = eos_pressure(ph, rhol, ul, vl, wl, L[4]);
  const double pr = eos_pressure(ph, rhor, ur, vr, wr, R[4]);
  const double al = eos_soundspeed(ph, rhol, pl);
  const double ar = eos_soundspeed(ph, rhor, pr);
  const double vnl = ul * fn[0] + vl * fn[1] + wl * fn[2];
  const double vnr = ur * fn[0] + vr * fn[1] + wr * fn[2];
  const double lambda = fmax(al, ar) + fmax(fabs(vnl), fabs(vnr));
  const double fl0 = L[0] * vnl, fr0 = R[0] * vnr;
  const double fl1 = L[1] * vnl + pl * fn[0], fr1 = R[1] * vnr + pr * fn[0];
  const double fl2 = L[2] * vnl + pl * fn[1], fr2 = R[2] * vnr + pr * fn[1];
  const double fl3 = L[3] * vnl + pl * fn[2], fr3 = R[3] * vnr + pr * fn[2];
  const double fl4 = (L[4] + pl) * vnl, fr4 = (R[4] + pr) * vnr;
  flx[0] = 0.5 * (fl0 + fr0 - lambda * (R[0] - L[0]));
  flx[1] = 0.5 * (fl1 + fr1 - lambda * (R[1] - L[1]));
  flx[2] = 0.5 * (fl2 + fr2 - lambda * (R[2] - L[2]));
  flx[3] = 0.5 * (fl3 + fr3 - lambda * (R[3] - L[3]));
  flx[4] = 0.5 * (fl4 + fr4 - lambda * (R[4] - L[4]));
}

__device__ __forceinline__ void riemann(const Phys& ph, const double* fn, const double* L,
                                        const double* R, double* flx)
{
  if (ph.flux == 1) flux_lf(ph, fn, L, R, flx);
  else flux_hllc(ph, fn, L, R, flx);
}

// Problem::solution (device functor per ProblemType):
// SodShocktube.cpp:28-78, SedovBlastwave.cpp:28-75, VorticalFlow.cpp:28-64,
// TaylorGreen.cpp:28-62 under src/PDE/CompFlow/Problem/
// NLEnergyGrowth.cpp:28-60
__device__ __forceinline__ double nleg_hx(const Phys& ph, double x, double y, double z)
{
  const double pi = 3.14159265358979323846;
  return cos(ph.betax * pi * x) * cos(ph.betay * pi * y) * cos(ph.betaz * pi * z);
}
__device__ __forceinline__ double nleg_ec(const Phys& ph, double t, double h, double p)
{
  return pow(-3.0 * (ph.ce + ph.kappa * h * h * t), p);
}

template <int PROB>
__device__ __forceinline__ void prob_solution(const Phys& ph, double x, double y, double z,
                                              double t, double* s)
{
  if constexpr (PROB == 6) {
    // RotatedSodShocktube.cpp:38-44: rotate back by -45 degrees about Z, Y, X (Vector.cpp:77-131)
    const double a = -45.0 * 3.14159265358979323846 / 180.0, ca = cos(a), sa = sin(a);
    double c0 = ca * x - sa * y, c1 = sa * x + ca * y, c2 = z;          // rotateZ
    { const double n0 = ca * c0 + sa * c2, n2 = -sa * c0 + ca * c2; c0 = n0; c2 = n2; }   // rotateY
    { const double n1 = ca * c1 - sa * c2, n2 = sa * c1 + ca * c2; c1 = n1; c2 = n2; }    // rotateX
    (void)c1; (void)c2;
    const bool l = c0 < 0.5;
    const double r = l ? 1.0 : 0.125, p = l ? 1.0 : 0.1;
    s[0] = r; s[1] = 0.0; s[2] = 0.0; s[3] = 0.0;
    s[4] = eos_totalenergy(ph, r, 0.0, 0.0, 0.0, p);
  } else if constexpr (PROB == 10) {
    // RayleighTaylor.cpp:28-62
    const double pi = 3.14159265358979323846;
    const double gx = ph.betax * x * x + ph.betay * y * y + ph.betaz * z * z;
    const double r = ph.r0 - gx, p = ph.p0 + ph.alpha * gx;
    const double ft = cos(ph.kappa * pi * t);
    const double u = ft * z * sin(pi * x), v = ft * z * cos(pi * y);
    const double w = ft * (-0.5 * pi * z * z * (cos(pi * x) - sin(pi * y)));
    s[0] = r; s[1] = r * u; s[2] = r * v; s[3] = r * w;
    s[4] = eos_totalenergy(ph, r, u, v, w, p);
  } else if constexpr (PROB == 7) {
    // NLEnergyGrowth.cpp:62-101
    const double gx = 1.0 - x * x - y * y - z * z;
    const double h = nleg_hx(ph, x, y, z);
    const double r = ph.r0 + exp(-ph.alpha * t) * gx;
    s[0] = r; s[1] = 0.0; s[2] = 0.0; s[3] = 0.0;
    s[4] = r * nleg_ec(ph, t, h, -1.0 / 3.0);
  } else if constexpr (PROB == 1) {
    const bool l = x < 0.5;
    const double r = l ? 1.0 : 0.125, p = l ? 1.0 : 0.1;
    s[0] = r; s[1] = 0.0; s[2] = 0.0; s[3] = 0.0;
    s[4] = eos_totalenergy(ph, r, 0.0, 0.0, 0.0, p);
  } else if constexpr (PROB == 2) {
    const double r = 1.0, p = ((x < 0.05) && (y < 0.05)) ? 783.4112 : 1.0e-6;
    s[0] = r; s[1] = 0.0; s[2] = 0.0; s[3] = 0.0;
    s[4] = eos_totalenergy(ph, r, 0.0, 0.0, 0.0, p);
  } else if constexpr (PROB == 3) {
    const double a = ph.alpha, b = ph.beta;
    const double ru = a * x - b * y, rv = b * x + a * y, rw = -2.0 * a * z;
    s[0] = 1.0; s[1] = ru; s[2] = rv; s[3] = rw;
    s[4] = (ru * ru + rv * rv + rw * rw) / 2.0 + (ph.p0 - 2.0 * a * a * z * z) / (ph.gamma - 1.0);
  } else if constexpr (PROB == 4) {
    const double pi = 3.14159265358979323846;
    const double r = 1.0;
    const double p = 10.0 + r / 4.0 * (cos(2.0 * pi * x) + cos(2.0 * pi * y));
    const double u = sin(pi * x) * cos(pi * y), v = -cos(pi * x) * sin(pi * y), w = 0.0;
    s[0] = r; s[1] = r * u; s[2] = r * v; s[3] = r * w;
    s[4] = eos_totalenergy(ph, r, u, v, w, p);
  } else {
    s[0] = s[1] = s[2] = s[3] = s[4] = 0.0;
  }
}

// Problem::src: VorticalFlow.cpp:80-115, TaylorGreen.cpp:77-90 (zero otherwise)
template <int PROB> constexpr bool prob_has_source() { return PROB == 3 || PROB == 4 || PROB == 7 || PROB == 10; }
// components whose source is identically zero (their projection is skipped)
template <int PROB> constexpr bool prob_src_is_zero(int c)
{
  return PROB == 3 ? (c == 0 || c == 3) : PROB == 4 ? c < 4 : !prob_has_source<PROB>();
}
template <int PROB>
__device__ __forceinline__ void prob_src(const Phys& ph, double x, double y, double z,
                                         double t, double* r)
{
  if constexpr (PROB == 10) {
    // RayleighTaylor.cpp:95-175
    const double pi = 3.14159265358979323846;
    const double a = ph.alpha, bx = ph.betax, by = ph.betay, bz = ph.betaz, kp = ph.kappa, g = ph.gamma;
    double s[5];
    prob_solution<10>(ph, x, y, z, t, s);
    const double rho = s[0], u = s[1] / s[0], v = s[2] / s[0], w = s[3] / s[0], E = s[4] / s[0];
    const double p = ph.p0 + a * (bx * x * x + by * y * y + bz * z * z);
    const double drdx[3] = { -2.0 * bx * x, -2.0 * by * y, -2.0 * bz * z };
    const double dpdx[3] = { 2.0 * a * bx * x, 2.0 * a * by * y, 2.0 * a * bz * z };
    const double ft = cos(kp * pi * t), st = sin(kp * pi * t);
    const double dudx[3] = { ft * pi * z * cos(pi * x), 0.0, ft * sin(pi * x) };
    const double dvdx[3] = { 0.0, -ft * pi * z * sin(pi * y), ft * cos(pi * y) };
    const double dwdx[3] = { ft * pi * 0.5 * pi * z * z * sin(pi * x), ft * pi * 0.5 * pi * z * z * cos(pi * y),
                             -ft * pi * z * (cos(pi * x) - sin(pi * y)) };
    const double dudt = -kp * pi * st * z * sin(pi * x);
    const double dvdt = -kp * pi * st * z * cos(pi * y);
    const double dwdt = kp * pi * st / 2 * pi * z * z * (cos(pi * x) - sin(pi * y));
    const double dedt = u * dudt + v * dvdt + w * dwdt;
    double dedx[3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
      dedx[d] = dpdx[d] / rho / (g - 1.0) - p / (g - 1.0) / rho / rho * drdx[d]
              + u * dudx[d] + v * dvdx[d] + w * dwdx[d];
    r[0] = u * drdx[0] + v * drdx[1] + w * drdx[2];
    r[1] = rho * dudt + u * r[0] + dpdx[0] + s[1] * dudx[0] + s[2] * dudx[1] + s[3] * dudx[2];
    r[2] = rho * dvdt + v * r[0] + dpdx[1] + s[1] * dvdx[0] + s[2] * dvdx[1] + s[3] * dvdx[2];
    r[3] = rho * dwdt + w * r[0] + dpdx[2] + s[1] * dwdx[0] + s[2] * dwdx[1] + s[3] * dwdx[2];
    r[4] = rho * dedt + E * r[0] + s[1] * dedx[0] + s[2] * dedx[1] + s[3] * dedx[2]
         + u * dpdx[0] + v * dpdx[1] + w * dpdx[2];
  } else if constexpr (PROB == 7) {
    // NLEnergyGrowth.cpp:124-190
    const double pi = 3.14159265358979323846;
    const double a = ph.alpha, bx = ph.betax, by = ph.betay, bz = ph.betaz, g = ph.gamma;
    const double gx = 1.0 - x * x - y * y - z * z;
    const double dg[3] = { -2.0 * x, -2.0 * y, -2.0 * z };
    const double h = nleg_hx(ph, x, y, z);
    const double dh[3] = { -bx * pi * sin(bx * pi * x) * cos(by * pi * y) * cos(bz * pi * z),
                           -by * pi * cos(bx * pi * x) * sin(by * pi * y) * cos(bz * pi * z),
                           -bz * pi * cos(bx * pi * x) * cos(by * pi * y) * sin(bz * pi * z) };
    const double ft = exp(-a * t), dfdt = -a * ft;
    const double rho = ph.r0 + ft * gx;
    const double drdt = gx * dfdt;
    const double ie = nleg_ec(ph, t, h, -1.0 / 3.0);
    const double ie4 = pow(ie, 4.0);
    const double dedt = ph.kappa * h * h * ie4;
    r[0] = drdt;
#pragma unroll
    for (int d = 0; d < 3; ++d)
      r[1 + d] = (g - 1.0) * (rho * (2.0 * ie4 * ph.kappa * h * dh[d] * t) + ie * (ft * dg[d]));
    r[4] = rho * dedt + ie * drdt;
  } else if constexpr (PROB == 3) {
    const double a = ph.alpha, b = ph.beta;
    double s[5];
    prob_solution<3>(ph, x, y, z, 0.0, s);
    r[0] = 0.0;
    r[1] = a * s[1] / s[0] - b * s[2] / s[0];
    r[2] = b * s[1] / s[0] + a * s[2] / s[0];
    r[3] = 0.0;
    r[4] = (r[1] * s[1] + r[2] * s[2]) / s[0] + 8.0 * a * a * a * z * z / (ph.gamma - 1.0);
  } else if constexpr (PROB == 4) {
    const double pi = 3.14159265358979323846;
    r[0] = r[1] = r[2] = r[3] = 0.0;
    r[4] = 3.0 * pi / 8.0 * (cos(3.0 * pi * x) * cos(pi * y) - cos(3.0 * pi * y) * cos(pi * x));
  } else {
    r[0] = r[1] = r[2] = r[3] = r[4] = 0.0;
  }
}

// BC state functions, src/PDE/CompFlow/DGCompFlow.hpp:649-701
template <int PROB>
__device__ __forceinline__ void bc_state(const Phys& ph, int bc, const double* ul, double x,
                                         double y, double z, double t, const double* fn,
                                         double* ur)
{
  if (bc == 1) {
    prob_solution<PROB>(ph, x, y, z, t, ur);
  } else if (bc == 2) {
    const double v1 = ul[1] / ul[0], v2 = ul[2] / ul[0], v3 = ul[3] / ul[0];
    const double vn = v1 * fn[0] + v2 * fn[1] + v3 * fn[2];
    ur[0] = ul[0];
    ur[1] = ur[0] * (v1 - 2.0 * vn * fn[0]);
    ur[2] = ur[0] * (v2 - 2.0 * vn * fn[1]);
    ur[3] = ur[0] * (v3 - 2.0 * vn * fn[2]);
    ur[4] = ul[4];
  } else {
    ur[0] = ul[0]; ur[1] = ul[1]; ur[2] = ul[2]; ur[3] = ul[3]; ur[4] = ul[4];
  }
}

// local face -> local nodes (src/Mesh/DerivedData.hpp:36), as selects so that
// no runtime-indexed array ends up in scratch
__device__ __forceinline__ int lpofa(int lf, int j)
{
  // {1,2,3},{2,0,3},{3,0,1},{0,2,1}
  const int packed = (lf == 0) ? 0x39 /*1,2,3*/ : (lf == 1) ? 0x32 /*2,0,3*/
                   : (lf == 2) ? 0x13 /*3,0,1*/ : 0x18 /*0,2,1*/;
  return (packed >> (2 * j)) & 3;
}

// reference coords in the neighbour of a face point with barycentric weights
// (s0,s1,s2) on the face's nodes, whose neighbour-local ids are in `code`
__device__ __forceinline__ void nbr_ref_coords(int code, double s0, double s1, double s2,
                                               double& xi, double& eta, double& zeta)
{
  const int m0 = code & 3, m1 = (code >> 2) & 3, m2 = (code >> 4) & 3;
  xi   = (m0 == 1 ? s0 : 0.0) + (m1 == 1 ? s1 : 0.0) + (m2 == 1 ? s2 : 0.0);
  eta  = (m0 == 2 ? s0 : 0.0) + (m1 == 2 ? s1 : 0.0) + (m2 == 2 ? s2 : 0.0);
  zeta = (m0 == 3 ? s0 : 0.0) + (m1 == 3 ? s1 : 0.0) + (m2 == 3 ? s2 : 0.0);
}

template <int NDOF>
__device__ __forceinline__ void load_dofs(const double* __restrict__ U, int /*stride*/, int e,
                                          double (&u)[NCOMP][NDOF])
{
  load_row<NCOMP * NDOF>(U, e, &u[0][0]);
}

template <int NDOF>
__device__ __forceinline__ void state_from(const double (&u)[NCOMP][NDOF], const double* B,
                                           double* s)
{
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) {
    double a = u[c][0];
#pragma unroll
    for (int k = 1; k < NDOF; ++k) a += u[c][k] * B[k];
    s[c] = a;
  }
}

// state of element `n` at a point with basis B, reading its row from HBM/L2
template <int NDOF>
__device__ __forceinline__ void state_gather(const double* __restrict__ U, int /*stride*/, int n,
                                             const double* B, double* s)
{
  double r[NCOMP][NDOF];
  load_row<NCOMP * NDOF>(U, n, &r[0][0]);
  state_from<NDOF>(r, B, s);
}

struct ElemGeom {
  double p[4][3];
};

__device__ __forceinline__ void load_geom(const DevMesh& m, int e, ElemGeom& g)
{
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = m.inpoel[(size_t)i * m.stride + e];
    g.p[i][0] = m.x[n]; g.p[i][1] = m.y[n]; g.p[i][2] = m.z[n];
  }
}

// physical coordinates of the point with weights (s0,s1,s2) on local face lf
__device__ __forceinline__ void face_point(const ElemGeom& g, int lf, double s0, double s1,
                                           double s2, double* P)
{
  // node weights of the 4 local nodes
  double w[4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
    w[n] = (lpofa(lf, 0) == n ? s0 : 0.0) + (lpofa(lf, 1) == n ? s1 : 0.0)
         + (lpofa(lf, 2) == n ? s2 : 0.0);
#pragma unroll
  for (int d = 0; d < 3; ++d)
    P[d] = g.p[0][d] * w[0] + g.p[1][d] * w[1] + g.p[2][d] * w[2] + g.p[3][d] * w[3];
}

// inverse Jacobian of the tet map, src/Base/Vector.cpp:155-197
__device__ __forceinline__ void inverse_jacobian(const ElemGeom& g, double (&ji)[3][3])
{
  const double (*v)[3] = g.p;
  const double bx = v[1][0] - v[0][0], by = v[1][1] - v[0][1], bz = v[1][2] - v[0][2];
  const double cx = v[2][0] - v[0][0], cy = v[2][1] - v[0][1], cz = v[2][2] - v[0][2];
  const double dx = v[3][0] - v[0][0], dy = v[3][1] - v[0][1], dz = v[3][2] - v[0][2];
  const double det = bx * (cy * dz - cz * dy) + by * (cz * dx - cx * dz) + bz * (cx * dy - cy * dx);
  const double id = 1.0 / det;
  ji[0][0] =  (cy * dz - dy * cz) * id;
  ji[1][0] = -(by * dz - dy * bz) * id;
  ji[2][0] =  (by * cz - cy * bz) * id;
  ji[0][1] = -(cx * dz - dx * cz) * id;
  ji[1][1] =  (bx * dz - dx * bz) * id;
  ji[2][1] = -(bx * cz - cx * bz) * id;
  ji[0][2] =  (cx * dy - dx * cy) * id;
  ji[1][2] = -(bx * dy - dx * by) * id;
  ji[2][2] =  (bx * cy - cx * by) * id;
}

// ------------------------------------------------------- fast fp64 helpers
// 1/x and sqrt(x) from the hardware seeds (v_rcp_f64 / v_rsq_f64) plus Newton
// steps: ~1 ulp, without the div_scale/div_fixup range handling of the full
// IEEE expansions (operands here are densities, pressures, wave-speed
// differences: far from the subnormal/overflow range).  NaN in -> NaN out.
__device__ __forceinline__ double fast_rcp(double x)
{
  double r = __builtin_amdgcn_rcp(x);
#if QDG_RCP_NR >= 1
  r = fma(fma(-x, r, 1.0), r, r);
#endif
#if QDG_RCP_NR >= 2
  r = fma(fma(-x, r, 1.0), r, r);
#endif
  return r;
}
__device__ __forceinline__ double fast_sqrt(double x)
{
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
#if QDG_SQRT_NR >= 2
  r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
#endif
  g = fma(fma(-g, g, x), h, g);
  return (x == 0.0) ? 0.0 : g;
}

// halo pack folded into a producing kernel's epilogue: device row d of the new state also goes to the
// send slab rows that carry it (DG::next / DG::lim pack m_u[tet] per neighbour, DG.cpp:1023-1031, 1262-1279)
template <int NPROP>
__device__ __forceinline__ void halo_fold_row(const DevMesh& m, int d, const double* row)
{
  if (!m.fold_slot || d < m.ninner || d >= m.nie) return;
  const int* sl = m.fold_slot + FOLD_SLOTS * (size_t)(d - m.ninner);
#pragma unroll 1
  for (int q = 0; q < FOLD_SLOTS; ++q) {
    const int j = sl[q];
    if (j < 0) break;
    store_row<NPROP>(m.fold_slab, j, row);
  }
}

// primitive quantities of one side of a face
struct Prim {
  double ir, p, a, vn;
};
__device__ __forceinline__ void primitives(const Phys& ph, const double* fn, const double* s, Prim& q)
{
  // p = (rhoE - |m|^2/(2 rho) - pc)(gamma-1) - pc,  a = sqrt(gamma (p+pc)/rho),  vn = (m.n)/rho
  q.ir = fast_rcp(s[0]);
  const double m2 = s[1] * s[1] + s[2] * s[2] + s[3] * s[3];
  q.p = (s[4] - 0.5 * m2 * q.ir - ph.pstiff) * (ph.gamma - 1.0) - ph.pstiff;
  q.a = fast_sqrt(ph.gamma * (q.p + ph.pstiff) * q.ir);
  q.vn = (s[1] * fn[0] + s[2] * fn[1] + s[3] * fn[2]) * q.ir;
}

// HLLC with precomputed primitives (same ladder as flux_hllc above)
__device__ __forceinline__ void flux_hllc_q(const double* fn, const double* L, const double* R,
                                            const Prim& ql, const Prim& qr, double* flx)
{
  const double rhol = L[0], rhor = R[0];
  const double rlr = fast_sqrt(rhor * ql.ir);
  const double irlr1 = fast_rcp(1.0 + rlr);
  const double vnroe = (qr.vn * rlr + ql.vn) * irlr1;
  const double aroe = (qr.a * rlr + ql.a) * irlr1;
  const double Sl = fmin(ql.vn - ql.a, vnroe - aroe);
  const double Sr = fmax(qr.vn + qr.a, vnroe + aroe);
  const double ml = rhol * (Sl - ql.vn), mr = rhor * (Sr - qr.vn);
  const double Sm = (mr * qr.vn - ml * ql.vn + ql.p - qr.p) * fast_rcp(mr - ml);
  const double pStar = rhol * (ql.vn - Sl) * (ql.vn - Sm) + ql.p;
  const bool c1 = Sl > 0.0;
  const bool c2 = !c1 && (Sl <= 0.0) && (Sm > 0.0);
  const bool c3 = !c1 && !c2 && (Sm <= 0.0) && (Sr >= 0.0);
  const bool left = c1 || c2;
  const bool star = c2 || c3;
  const double S = left ? Sl : Sr;
  const double vn = left ? ql.vn : qr.vn;
  const double p = left ? ql.p : qr.p;
  const double u0 = left ? L[0] : R[0], u1 = left ? L[1] : R[1], u2 = left ? L[2] : R[2],
               u3 = left ? L[3] : R[3], u4 = left ? L[4] : R[4];
  // star:  F = U* Sm + (0, p* n, p* Sm),  U* = ((S-vn) U + (0, (p*-p) n, p* Sm - p vn)) / (S-Sm)
  // plain: F = U vn + (0, p n, p vn)          -> one expression with selected factors
  const double id = star ? fast_rcp(S - Sm) : 1.0;
  const double sv = star ? (S - vn) * id * Sm : vn;         // factor on U
  const double dp = star ? (pStar - p) * id * Sm + pStar : p; // factor on n
  const double e4 = star ? ((pStar * Sm - p * vn) * id + pStar) * Sm : p * vn;
  flx[0] = sv * u0;
  flx[1] = sv * u1 + dp * fn[0];
  flx[2] = sv * u2 + dp * fn[1];
  flx[3] = sv * u3 + dp * fn[2];
  flx[4] = sv * u4 + e4;
}

__device__ __forceinline__ void flux_lf_q(const double* fn, const double* L, const double* R,
                                          const Prim& ql, const Prim& qr, double* flx)
{
  const double lambda = fmax(ql.a, qr.a) + fmax(fabs(ql.vn), fabs(qr.vn));
  const double fl[5] = { L[0] * ql.vn, L[1] * ql.vn + ql.p * fn[0], L[2] * ql.vn + ql.p * fn[1],
                         L[3] * ql.vn + ql.p * fn[2], (L[4] + ql.p) * ql.vn };
  const double fr[5] = { R[0] * qr.vn, R[1] * qr.vn + qr.p * fn[0], R[2] * qr.vn + qr.p * fn[1],
                         R[3] * qr.vn + qr.p * fn[2], (R[4] + qr.p) * qr.vn };
#pragma unroll
  for (int c = 0; c < 5; ++c) flx[c] = 0.5 * (fl[c] + fr[c] - lambda * (R[c] - L[c]));
}
// HLLC in the OWN tet's frame (left' = own, right' = neighbour, n' = the own tet's outward
// normal).  For a face whose stored left tet is the neighbour this is the mirror image of the
// reference's evaluation (Sl' = -Sr, Sm' = -Sm, Sr' = -Sl), so the reference's ladder
// (HLLC.hpp:93-124) is applied in its mirrored form: the same four fluxes and the same
// fall-through of a NaN wave speed to the STORED right state.
__device__ __forceinline__ void flux_hllc_own(const double* fn, const double* so, const double* sn,
                                              const Prim& qo, const Prim& qn, bool own_left, double* flx)
{
  const double rlr = fast_sqrt(sn[0] * qo.ir);
  const double irlr1 = fast_rcp(1.0 + rlr);
  const double vnroe = (qn.vn * rlr + qo.vn) * irlr1;
  const double aroe = (qn.a * rlr + qo.a) * irlr1;
  const double Sl = fmin(qo.vn - qo.a, vnroe - aroe);
  const double Sr = fmax(qn.vn + qn.a, vnroe + aroe);
  const double ml = so[0] * (Sl - qo.vn), mr = sn[0] * (Sr - qn.vn);
  const double Sm = (mr * qn.vn - ml * qo.vn + qo.p - qn.p) * fast_rcp(mr - ml);
  const double pStar = so[0] * (qo.vn - Sl) * (qo.vn - Sm) + qo.p;
  const bool c1 = Sl > 0.0;
  const bool c2 = !c1 && (Sl <= 0.0) && (Sm > 0.0);
  const bool c3 = !c1 && !c2 && (Sm <= 0.0) && (Sr >= 0.0);
  const bool m1 = Sr < 0.0;
  const bool m2 = !m1 && (Sr >= 0.0) && (Sm < 0.0);
  const bool m3 = !m1 && !m2 && (Sm >= 0.0) && (Sl <= 0.0);
  const bool left = own_left ? (c1 || c2) : !(m1 || m2);
  const bool star = own_left ? (c2 || c3) : (m2 || m3);
  const double S = left ? Sl : Sr;
  const double vn = left ? qo.vn : qn.vn;
  const double p = left ? qo.p : qn.p;
  const double u0 = left ? so[0] : sn[0], u1 = left ? so[1] : sn[1], u2 = left ? so[2] : sn[2],
               u3 = left ? so[3] : sn[3], u4 = left ? so[4] : sn[4];
  const double id = star ? fast_rcp(S - Sm) : 1.0;
  const double sv = star ? (S - vn) * id * Sm : vn;
  const double dp = star ? (pStar - p) * id * Sm + pStar : p;
  const double e4 = star ? ((pStar * Sm - p * vn) * id + pStar) * Sm : p * vn;
  flx[0] = sv * u0;
  flx[1] = sv * u1 + dp * fn[0];
  flx[2] = sv * u2 + dp * fn[1];
  flx[3] = sv * u3 + dp * fn[2];
  flx[4] = sv * u4 + e4;
}

#define QDG_DISPATCH_PDG(m, CALL)                              \
  do {                                                         \
    if ((m).ndofel) { constexpr bool G = true; CALL; }         \
    else { constexpr bool G = false; CALL; }                   \
  } while (0)

#define QDG_DISPATCH_NDOF(ndof, CALL)          \
  do {                                          \
    if ((ndof) == 1) { constexpr int N = 1; CALL; }       \
    else if ((ndof) == 4) { constexpr int N = 4; CALL; }  \
    else { constexpr int N = 10; CALL; }                  \
  } while (0)

#define QDG_DISPATCH_PROB(prob, CALL)                      \
  do {                                                      \
    switch (prob) {                                         \
      case 1: { constexpr int P = 1; CALL; } break;         \
      case 2: { constexpr int P = 2; CALL; } break;         \
      case 3: { constexpr int P = 3; CALL; } break;         \
      case 4: { constexpr int P = 4; CALL; } break;         \
      case 6: { constexpr int P = 6; CALL; } break;         \
      case 7: { constexpr int P = 7; CALL; } break;         \
      case 10: { constexpr int P = 10; CALL; } break;       \
      default: { constexpr int P = 0; CALL; } break;        \
    }                                                       \
  } while (0)

static inline int nblk(int n, int b) { return (n + b - 1) / b; }

// P1 basis functions 1..3 at reference vertex v (tile kernels, fused update + limiter)
__device__ __forceinline__ void vertex_basis(int v, double& b1, double& b2, double& b3)
{
  // B1 = 2xi+eta+zeta-1, B2 = 3eta+zeta-1, B3 = 4zeta-1 at reference vertex v
  b1 = (v == 0) ? -1.0 : (v == 1) ? 1.0 : 0.0;
  b2 = (v == 2) ? 2.0 : (v == 3) ? 0.0 : -1.0;
  b3 = (v == 3) ? 3.0 : -1.0;
}

// fills THIS translation unit's constant tables
static inline hipError_t upload_tables_here(const Tables<1>& t1, const Tables<4>& t4, const Tables<10>& t10,
                                            const QuadTet* qinit, const QuadTet* qdiag)
{
  hipError_t e;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_tab1), &t1, sizeof(t1))) != hipSuccess) return e;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_tab4), &t4, sizeof(t4))) != hipSuccess) return e;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_tab10), &t10, sizeof(t10))) != hipSuccess) return e;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_qinit), qinit, 3 * sizeof(QuadTet))) != hipSuccess) return e;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_qdiag), qdiag, 3 * sizeof(QuadTet))) != hipSuccess) return e;
  return hipSuccess;
}

}  // namespace qdg
