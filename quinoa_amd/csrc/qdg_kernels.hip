// qdg_kernels.hip -- hand-written gfx950 (MI355X, CDNA4) kernels of the DG
// compressible-flow path.  Fields are element-major rows in the Morton-ordered device
// numbering (qdg_device.hpp), so a face-neighbour gather is one contiguous row served by
// the XCD's L2; rows are written back through LDS as coalesced wave stores.
//
// Right-hand side, by scheme order:
//   DG-P1  k_rhs_p1v / k_rhs_p1t  tile / face-task kernels: a workgroup owns 248 consecutive
//          rows, keeps their vertex states and flux accumulators in LDS and evaluates every
//          face between two tets of the tile ONCE (one lane per face task, ds_add_f64 into
//          both tets); k_rhs_p1 is the element-centric, bitwise reproducible form.
//   DG-P2  k_rhs_p2s  a LANE PAIR per tet (modes split 5/5, partial sums exchanged by DPP,
//          basis values from LDS tables, 2 waves per SIMD); k_rhs_p2 one lane per tet with a
//          face's six Gauss points batched; k_rhs<10> the round-1 reference form.
//   DG-P0  k_rhs<1>.
// The element-centric kernels evaluate a face's Riemann flux from both of its tets with the
// same expression (stored orientation, or the own tet's frame with the mirrored HLLC ladder),
// so no atomics are needed and R is written exactly once.  Stage 0 fuses the CFL time-step sum
// of dg::CompFlow::dt into the face loop; whenever dt is known before the launch the SSP-RK3
// update is fused in as well and R never goes to memory.
//
// Reference coordinates of face Gauss points are constant tables: on a
// straight-sided tet they depend only on the local face id (own side) and on
// the 3-node permutation code `finfo` (neighbour side), which removes the six
// tk::Jacobian evaluations per Gauss point of Surface.cpp:159-166.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include "qdg_device.hpp"
#include "qdg_kernels.hpp"
#include "qdg_tables.hpp"

#ifndef QDG_RCP_NR
#define QDG_RCP_NR 1    // Newton steps after v_rcp_f64 (1 step: R agrees with the fp64-division CPU result to 1e-15)
#endif
#ifndef QDG_SQRT_NR
#define QDG_SQRT_NR 1   // Goldschmidt steps after v_rsq_f64 (plus one residual correction)
#endif
#ifndef QDG_TILE_WAVES
#define QDG_TILE_WAVES 2
#endif
#ifndef QDG_TILE_GP_UNROLL
#define QDG_TILE_GP_UNROLL 1
#endif
#ifndef QDG_TILE_GP_SERIAL
#define QDG_TILE_GP_SERIAL 1
#endif
#ifndef QDG_P2_ILP
#define QDG_P2_ILP 0
#endif
#ifndef QDG_P1_WAVES
#define QDG_P1_WAVES 2   // waves per SIMD the DG-P1 RHS kernel is register-budgeted for
#endif

namespace qdg {

__constant__ Tables<1> c_tab1;
__constant__ Tables<4> c_tab4;
__constant__ Tables<10> c_tab10;
__constant__ QuadTet c_qinit[3];   // NGinit rule per order index
__constant__ QuadTet c_qdiag[3];   // NGdiag rule per order index
__device__ P2Split g_p2s;          // copied to LDS by every workgroup of k_rhs_p2s

#ifdef QDG_STAMPS
// diagnostic build only: per-segment cycle sums of the P1 RHS kernel (lane 0 of
// every wave adds its s_memtime differences); never part of a timed build
__device__ unsigned long long g_stamp[16];
#define STAMP(i)                                                               \
  do {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                         \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                        \
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_stamp[i], t_ - tprev_);          \
    tprev_ = __builtin_amdgcn_s_memtime();                                     \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
#define STAMP_INIT unsigned long long tprev_ = __builtin_amdgcn_s_memtime()
#else
#define STAMP(i) do {} while (0)
#define STAMP_INIT do {} while (0)
#endif

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

// ------------------------------------------------------------- RHS kernel
// dg::CompFlow::rhs (src/PDE/CompFlow/DGCompFlow.hpp:130-195) for interior
// tets: surfInt + bndSurfInt (per local face), volInt, srcInt.
// Rows are read ONCE into registers: the tet's own row before the face loop, a
// neighbour's row once per face (P2: 3 x 100 VGPRs of rows and accumulators --
// the kernel is built for one wave per SIMD, where a wave may hold 512 registers).
// WITH_DT (stage 0 with a CFL time step): dg::CompFlow::dt needs |vn|+a of both
// sides at every face Gauss point -- what the Riemann solver has just computed;
// the per-workgroup minimum of vol/delt goes to blockmin (k_dt_final finishes).
// MODE 0: R = rhs(U); 1: + CFL time step (stage 0); 2: the SSP-RK3 update fused in,
// R <- a*Un + b*(U + dt*rhs/L) (stages 1, 2: dt is known, the RHS never goes to memory)
template <int NDOF, int PROB, int MODE>
__global__ __launch_bounds__(256, (NDOF > 4 ? 1 : 2)) void k_rhs(DevMesh m, Phys ph, double t,
                                             const double* __restrict__ U,
                                             double* __restrict__ R,
                                             double* __restrict__ blockmin,
                                             double rk_a, double rk_b,
                                             const double* __restrict__ dtp,
                                             const double* __restrict__ Un)
{
  constexpr bool WITH_DT = MODE == 1, FUSE_RK = MODE == 2;
  const int e0 = xcd_tile(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  const bool active = e0 < m.nie;
  if (!WITH_DT && !active) return;
  const int e = active ? e0 : m.nie - 1;       // WITH_DT: every lane reaches the reduction
  double delt = 0.0;
  const Tables<NDOF>& T = tab<NDOF>();
  constexpr int NGF = Tables<NDOF>::NGF, NGV = Tables<NDOF>::NGV;
  const int stride = m.stride;

  double acc[NCOMP][NDOF], u[NCOMP][NDOF];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c)
#pragma unroll
    for (int k = 0; k < NDOF; ++k) acc[c][k] = 0.0;
  load_dofs<NDOF>(U, stride, e, u);

  ElemGeom g;
  load_geom(m, e, g);

  // ---- faces ------------------------------------------------------------
#pragma unroll 1
  for (int lf = 0; lf < 4; ++lf) {
    const int nb = m.nbr[(size_t)lf * stride + e];
    // boundary face without a BC: no flux, but dg::CompFlow::dt still counts it
    // (its face loop runs over all faces, DGCompFlow.hpp:226)
    if (nb == -1 && !WITH_DT) continue;
    const double wsel = (nb == -1) ? 0.0 : 1.0;
    const int info = m.finfo[(size_t)lf * stride + e];
    const int f = m.fid[(size_t)lf * stride + e];
    const double area = m.farea[f];
    const double fn[3] = { m.fnx[f], m.fny[f], m.fnz[f] };
    const bool own_left = (info >> 6) & 1;
    double un[NCOMP][NDOF];
    if (nb >= 0) load_dofs<NDOF>(U, stride, nb, un);
#pragma unroll 1
    for (int ig = 0; ig < NGF; ++ig) {
      const double s0 = T.fs[ig][0], s1 = T.fs[ig][1], s2 = T.fs[ig][2];
      double so[NCOMP], sn[NCOMP], fl[NCOMP];
      state_from<NDOF>(u, T.fB[lf][ig], so);
      if (nb >= 0) {
        double xi, eta, zeta, Bn[NDOF];
        nbr_ref_coords(info, s0, s1, s2, xi, eta, zeta);
        eval_basis<NDOF>(xi, eta, zeta, Bn);
        state_from<NDOF>(un, Bn, sn);
      } else {
        double P[3];
        face_point(g, lf, s0, s1, s2, P);
        bc_state<PROB>(ph, -nb - 1, so, P[0], P[1], P[2], t, fn, sn);
      }
      double L[NCOMP], Rr[NCOMP];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) { L[c] = own_left ? so[c] : sn[c]; Rr[c] = own_left ? sn[c] : so[c]; }
      Prim ql, qr;
      primitives(ph, fn, L, ql);
      primitives(ph, fn, Rr, qr);
      const double wq = T.fw[ig] * area;
      if (WITH_DT) {
        // std::max(dSV_l, dSV_r) as (a < b) ? b : a in (face-left, face-right) order
        const double dl = wq * (fabs(ql.vn) + ql.a);
        const double dr = (nb < 0) ? 0.0 : wq * (fabs(qr.vn) + qr.a);
        delt += (dl < dr) ? dr : dl;
      }
      if (ph.flux == 1) flux_lf_q(fn, L, Rr, ql, qr, fl);
      else flux_hllc_q(fn, L, Rr, ql, qr, fl);
      const double wt = (own_left ? -1.0 : 1.0) * wq * wsel;
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const double wf = wt * fl[c];
        acc[c][0] += wf;
#pragma unroll
        for (int k = 1; k < NDOF; ++k) acc[c][k] += wf * T.fB[lf][ig][k];
      }
    }
  }

  const double vol = m.vol[e];

  // ---- volume integral, src/PDE/Integrate/Volume.cpp:20-168 -------------
  if constexpr (NDOF > 1) {
    double ji[3][3];
    inverse_jacobian(g, ji);
#pragma unroll 1
    for (int ig = 0; ig < NGV; ++ig) {
      double s[NCOMP];
      state_from<NDOF>(u, T.vB[ig], s);
      const double ir = fast_rcp(s[0]);
      const double u = s[1] * ir, v = s[2] * ir, w = s[3] * ir;
      const double p = eos_pressure(ph, s[0], u, v, w, s[4]);
      const double wt = T.vw[ig] * vol;
      // Euler flux F[c][d], src/PDE/CompFlow/DGCompFlow.hpp:599-635
      const double F[NCOMP][3] = {
        { s[1], s[2], s[3] },
        { s[1] * u + p, s[2] * u, s[3] * u },
        { s[1] * v, s[2] * v + p, s[3] * v },
        { s[1] * w, s[2] * w, s[3] * w + p },
        { u * (s[4] + p), v * (s[4] + p), w * (s[4] + p) } };
#pragma unroll
      for (int k = 1; k < NDOF; ++k) {
        // dB_k/dx_d = sum_j dB_k/dxi_j * jacInv[j][d]   (Basis.cpp:77-265)
        const double g0 = T.vdB[ig][0][k], g1 = T.vdB[ig][1][k], g2 = T.vdB[ig][2][k];
        const double dx = g0 * ji[0][0] + g1 * ji[1][0] + g2 * ji[2][0];
        const double dy = g0 * ji[0][1] + g1 * ji[1][1] + g2 * ji[2][1];
        const double dz = g0 * ji[0][2] + g1 * ji[1][2] + g2 * ji[2][2];
#pragma unroll
        for (int c = 0; c < NCOMP; ++c)
          acc[c][k] += wt * (F[c][0] * dx + F[c][1] * dy + F[c][2] * dz);
      }
    }
  }

  // ---- source integral, src/PDE/Integrate/Source.cpp:21-141 -------------
  if constexpr (prob_has_source<PROB>()) {
#pragma unroll 1
    for (int ig = 0; ig < NGV; ++ig) {
      const double xi = T.vc[ig][0], eta = T.vc[ig][1], zeta = T.vc[ig][2];
      const double w0 = 1.0 - xi - eta - zeta;
      double P[3], s[NCOMP];
#pragma unroll
      for (int d = 0; d < 3; ++d)
        P[d] = g.p[0][d] * w0 + g.p[1][d] * xi + g.p[2][d] * eta + g.p[3][d] * zeta;
      prob_src<PROB>(ph, P[0], P[1], P[2], t, s);
      const double wt = T.vw[ig] * vol;
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const double ws = wt * s[c];
        acc[c][0] += ws;
#pragma unroll
        for (int k = 1; k < NDOF; ++k) acc[c][k] += ws * T.vB[ig][k];
      }
    }
  }

  if constexpr (FUSE_RK) {
    constexpr double imf[10] = { 1.0, 10.0, 10.0 / 3.0, 5.0 / 3.0, 35.0, 21.0, 14.0, 7.0,
                                 14.0 / 3.0, 7.0 / 3.0 };
    const double dtv = dtp[0] / m.vol[e];
    // Un row streamed component by component (the own row u is still in registers)
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      double un[NDOF];
#pragma unroll
      for (int k = 0; k < NDOF; ++k) un[k] = Un[(size_t)e * (NCOMP * NDOF) + c * NDOF + k];
#pragma unroll
      for (int k = 0; k < NDOF; ++k)
        acc[c][k] = rk_a * un[k] + rk_b * (u[c][k] + dtv * imf[k] * acc[c][k]);
    }
  }
  if (active) store_row<NCOMP * NDOF>(R, e, &acc[0][0]);
  if (WITH_DT) {
    double dte = active ? m.vol[e] / delt : DBL_MAX;
    for (int off = 32; off > 0; off >>= 1) dte = fmin(dte, __shfl_down(dte, off, 64));
    __shared__ double wmin[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wmin[wv] = dte;
    __syncthreads();
    if (threadIdx.x == 0)
      blockmin[blockIdx.x] = fmin(fmin(wmin[0], wmin[1]), fmin(wmin[2], wmin[3]));
  }
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

// ------------------------------------------------- DG-P2 RHS, face-batched form
// Same algorithm and MODEs as k_rhs<10>, one lane per tet, organised around what the issue
// slots of the generic kernel were spent on (its rows sit in accumulation registers at one wave
// per SIMD, and every Gauss point read all 150 doubles of u, the neighbour row and the
// accumulators through v_accvgpr moves -- a third of its instructions):
//  * a face's six Gauss points are handled together: the neighbour row is consumed into the six
//    neighbour states while it arrives, the own states are formed mode by mode (each mode of u
//    is read once per face, the basis values are scalar loads), the six fluxes are computed in
//    the own tet's frame (no left/right swaps), and the accumulators are visited once per face
//    with the six weighted fluxes of a component;
//  * the volume term contracts the Euler flux with the inverse Jacobian first
//    (G[c][j] = sum_d F[c][d] J^-1[j][d], 45 FMAs) and then with the reference gradients of the
//    basis (3 FMAs per accumulator) instead of forming dB/dx per mode and point;
//  * the new rows leave through LDS as coalesced wave stores.
template <int PROB, int MODE>
__global__ __launch_bounds__(256, 1) void k_rhs_p2(DevMesh m, Phys ph, double t,
                                                   const double* __restrict__ U,
                                                   double* __restrict__ R,
                                                   double* __restrict__ blockmin,
                                                   double rk_a, double rk_b,
                                                   const double* __restrict__ dtp,
                                                   const double* __restrict__ Un)
{
  constexpr int NDOF = 10, NGF = 6, NGV = 11, NPROP = NCOMP * NDOF;
  constexpr bool WITH_DT = MODE == 1, FUSE_RK = MODE == 2;
  __shared__ __attribute__((aligned(16))) double stage[256 * NPROP];
  const Tables<10>& T = c_tab10;
  const int blk = xcd_tile(blockIdx.x, gridDim.x);
  const int e0 = blk * 256 + threadIdx.x;
  const bool active = e0 < m.nie;
  const int e = active ? e0 : m.nie - 1;       // every lane runs to the barriers
  const int stride = m.stride;
  double delt = 0.0;

  double acc[NCOMP][NDOF], u[NCOMP][NDOF];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c)
#pragma unroll
    for (int k = 0; k < NDOF; ++k) acc[c][k] = 0.0;
  load_row<NPROP>(U, e, &u[0][0]);
  ElemGeom g;
  load_geom(m, e, g);

  // ---- faces ------------------------------------------------------------
#pragma unroll 1
  for (int lf = 0; lf < 4; ++lf) {
    const int nb = m.nbr[(size_t)lf * stride + e];
    if (nb == -1 && !WITH_DT) continue;       // boundary face without a BC (dt still counts it)
    const int info = m.finfo[(size_t)lf * stride + e];
    const int f = m.fid[(size_t)lf * stride + e];
    double gq[4];
    load_row<4>(m.fgeo, f, gq);
    const bool own_left = (info >> 6) & 1;
    const double area = gq[0];
    const double osg = own_left ? 1.0 : -1.0;
    const double fn[3] = { osg * gq[1], osg * gq[2], osg * gq[3] };

    double so[NGF][NCOMP], sn[NGF][NCOMP];
    if (nb >= 0) {
      double un[NCOMP][NDOF];
      load_row<NPROP>(U, nb, &un[0][0]);
#pragma unroll
      for (int ig = 0; ig < NGF; ++ig) {
        double xi, eta, zeta, Bn[NDOF];
        nbr_ref_coords(info, T.fs[ig][0], T.fs[ig][1], T.fs[ig][2], xi, eta, zeta);
        eval_basis<NDOF>(xi, eta, zeta, Bn);
        state_from<NDOF>(un, Bn, sn[ig]);
      }
    }
    // own states, mode by mode
#pragma unroll
    for (int c = 0; c < NCOMP; ++c)
#pragma unroll
      for (int ig = 0; ig < NGF; ++ig) so[ig][c] = u[c][0];
#pragma unroll
    for (int k = 1; k < NDOF; ++k)
#pragma unroll
      for (int c = 0; c < NCOMP; ++c)
#pragma unroll
        for (int ig = 0; ig < NGF; ++ig) so[ig][c] = fma(u[c][k], T.fB[lf][ig][k], so[ig][c]);
    if (nb < 0) {
#pragma unroll 1
      for (int ig = 0; ig < NGF; ++ig) {
        // (runtime index on so/sn would go to scratch: select the point's states)
        double P[3], sl[NCOMP], sr[NCOMP];
        face_point(g, lf, T.fs[ig][0], T.fs[ig][1], T.fs[ig][2], P);
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
          sl[c] = so[0][c];
#pragma unroll
          for (int j = 1; j < NGF; ++j) sl[c] = (ig == j) ? so[j][c] : sl[c];
        }
        bc_state<PROB>(ph, -nb - 1, sl, P[0], P[1], P[2], t, fn, sr);
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
#pragma unroll
          for (int j = 0; j < NGF; ++j) sn[j][c] = (ig == j) ? sr[c] : sn[j][c];
        }
      }
    }
    // fluxes in the own frame, weighted: the own tet loses what leaves through the face.
    // The six points are independent: one straight-line block for all of them lets the
    // scheduler interleave their dependency chains (one wave per SIMD has nothing else to
    // hide the fp64 latency with).
    const double wsel = (nb == -1) ? 0.0 : 1.0;
    const bool lf_flux = ph.flux == 1;
    double wq[NGF];
#pragma unroll
    for (int ig = 0; ig < NGF; ++ig) wq[ig] = T.fw[ig] * area;
    if (!lf_flux) {
#pragma unroll
      for (int ig = 0; ig < NGF; ++ig) {
        Prim qo, qn;
        primitives(ph, fn, so[ig], qo);
        primitives(ph, fn, sn[ig], qn);
        if (WITH_DT) {
          // std::max(dSV_l, dSV_r) as (a < b) ? b : a in STORED (left, right) order
          const double d_o = wq[ig] * (fabs(qo.vn) + qo.a);
          const double d_n = (nb < 0) ? 0.0 : wq[ig] * (fabs(qn.vn) + qn.a);
          const bool take_n = own_left ? (d_o < d_n) : !(d_n < d_o);
          delt += take_n ? d_n : d_o;
        }
        double fl[NCOMP];
        flux_hllc_own(fn, so[ig], sn[ig], qo, qn, own_left, fl);
        const double wt = -wq[ig] * wsel;
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) so[ig][c] = wt * fl[c];
#if !QDG_P2_ILP
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    } else {
#pragma unroll
      for (int ig = 0; ig < NGF; ++ig) {
        Prim qo, qn;
        primitives(ph, fn, so[ig], qo);
        primitives(ph, fn, sn[ig], qn);
        if (WITH_DT) {
          const double d_o = wq[ig] * (fabs(qo.vn) + qo.a);
          const double d_n = (nb < 0) ? 0.0 : wq[ig] * (fabs(qn.vn) + qn.a);
          const bool take_n = own_left ? (d_o < d_n) : !(d_n < d_o);
          delt += take_n ? d_n : d_o;
        }
        double fl[NCOMP];
        flux_lf_q(fn, so[ig], sn[ig], qo, qn, fl);
        const double wt = -wq[ig] * wsel;
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) so[ig][c] = wt * fl[c];
      }
    }
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      acc[c][0] += ((so[0][c] + so[1][c]) + (so[2][c] + so[3][c])) + (so[4][c] + so[5][c]);
#pragma unroll
      for (int k = 1; k < NDOF; ++k) {
        double a = acc[c][k];
#pragma unroll
        for (int ig = 0; ig < NGF; ++ig) a = fma(so[ig][c], T.fB[lf][ig][k], a);
        acc[c][k] = a;
      }
    }
  }

  const double vol = m.vol[e];

  // ---- volume integral, src/PDE/Integrate/Volume.cpp:20-168 -------------
  {
    double ji[3][3];
    inverse_jacobian(g, ji);
#if QDG_P2_ILP >= 2
#pragma unroll 2
#else
#pragma unroll 1
#endif
    for (int ig = 0; ig < NGV; ++ig) {
      double s[NCOMP];
      state_from<NDOF>(u, T.vB[ig], s);
      const double ir = fast_rcp(s[0]);
      const double uu = s[1] * ir, vv = s[2] * ir, ww = s[3] * ir;
      const double p = eos_pressure(ph, s[0], uu, vv, ww, s[4]);
      const double wt = T.vw[ig] * vol;
      const double h = s[4] + p;
      // Euler flux F[c][d], src/PDE/CompFlow/DGCompFlow.hpp:599-635
      const double F[NCOMP][3] = {
        { s[1], s[2], s[3] },
        { s[1] * uu + p, s[2] * uu, s[3] * uu },
        { s[1] * vv, s[2] * vv + p, s[3] * vv },
        { s[1] * ww, s[2] * ww, s[3] * ww + p },
        { uu * h, vv * h, ww * h } };
      // dB_k/dx_d = sum_j dB_k/dxi_j jacInv[j][d] (Basis.cpp:77-265): contract F with jacInv first
      double G[NCOMP][3];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          G[c][j] = wt * (F[c][0] * ji[j][0] + F[c][1] * ji[j][1] + F[c][2] * ji[j][2]);
#pragma unroll
      for (int k = 1; k < NDOF; ++k) {
        const double g0 = T.vdB[ig][0][k], g1 = T.vdB[ig][1][k], g2 = T.vdB[ig][2][k];
#pragma unroll
        for (int c = 0; c < NCOMP; ++c)
          acc[c][k] += G[c][0] * g0 + G[c][1] * g1 + G[c][2] * g2;
      }
    }
  }

  // ---- source integral, src/PDE/Integrate/Source.cpp:21-141 -------------
  if constexpr (prob_has_source<PROB>()) {
#pragma unroll 1
    for (int ig = 0; ig < NGV; ++ig) {
      const double xi = T.vc[ig][0], eta = T.vc[ig][1], zeta = T.vc[ig][2];
      const double w0 = 1.0 - xi - eta - zeta;
      double P[3], s[NCOMP];
#pragma unroll
      for (int d = 0; d < 3; ++d)
        P[d] = g.p[0][d] * w0 + g.p[1][d] * xi + g.p[2][d] * eta + g.p[3][d] * zeta;
      prob_src<PROB>(ph, P[0], P[1], P[2], t, s);
      const double wt = T.vw[ig] * vol;
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const double ws = wt * s[c];
        acc[c][0] += ws;
#pragma unroll
        for (int k = 1; k < NDOF; ++k) acc[c][k] += ws * T.vB[ig][k];
      }
    }
  }

  if constexpr (FUSE_RK) {
    constexpr double imf[10] = { 1.0, 10.0, 10.0 / 3.0, 5.0 / 3.0, 35.0, 21.0, 14.0, 7.0,
                                 14.0 / 3.0, 7.0 / 3.0 };
    const double dtv = dtp[0] / vol;
    // Un row streamed component by component (the own row u is still in registers)
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      double un[NDOF];
#pragma unroll
      for (int k = 0; k < NDOF; ++k) un[k] = Un[(size_t)e * NPROP + c * NDOF + k];
#pragma unroll
      for (int k = 0; k < NDOF; ++k)
        acc[c][k] = rk_a * un[k] + rk_b * (u[c][k] + dtv * imf[k] * acc[c][k]);
    }
  }
  // rows out, coalesced (see k_rhs_p1v): deposit in LDS, leave as 1-KiB wave stores
  {
    double2* row = reinterpret_cast<double2*>(stage + (size_t)threadIdx.x * NPROP);
#pragma unroll
    for (int j = 0; j < NPROP / 2; ++j) row[j] = make_double2((&acc[0][0])[2 * j], (&acc[0][0])[2 * j + 1]);
    __syncthreads();
    const int r0 = blk * 256;
    const int nrow = (m.nie - r0 < 256) ? m.nie - r0 : 256;
    const double2* src = reinterpret_cast<const double2*>(stage);
    double2* dst = reinterpret_cast<double2*>(R + (size_t)r0 * NPROP);
    const int nvalid = nrow * (NPROP / 2);
#pragma unroll 5
    for (int j = 0; j < NPROP / 2; ++j) {
      const int i = j * 256 + threadIdx.x;
      if (i < nvalid) dst[i] = src[i];
    }
  }
  if (WITH_DT) {
    double dte = active ? vol / delt : DBL_MAX;
    for (int off = 32; off > 0; off >>= 1) dte = fmin(dte, __shfl_down(dte, off, 64));
    __shared__ double wmin[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wmin[wv] = dte;
    __syncthreads();
    if (threadIdx.x == 0)
      blockmin[blockIdx.x] = fmin(fmin(wmin[0], wmin[1]), fmin(wmin[2], wmin[3]));
  }
}

// ------------------------------------------------- DG-P2 RHS, two lanes per tet
// The one-lane-per-tet forms above need the tet's row, its accumulators and a neighbour row
// (3 x 100 registers) and run at ONE wave per SIMD, where the vector unit idles 44 % of the
// time (profiles/r02_cfg3_nx55_pmc_per_launch.json: fp64 latency and memory waits with nothing
// to switch to).  Here a tet is worked on by a PAIR of adjacent lanes; lane half h owns the
// modes k in [5h, 5h+5) of every row: 50 registers each for u, the accumulators and a neighbour
// half row -> 2 waves per SIMD.
//  * a state at a point is the sum of the two lanes' partial sums over their modes; the pair
//    exchanges partial sums with one DPP quad_perm(1,0,3,2) move per 32-bit half;
//  * the six Gauss points of a face are taken as "slots": lane half h maps slot s to point
//    (3h + s) mod 6, so each lane's slots 0-2 are the points whose fluxes it computes and its
//    slots 3-5 are its partner's -- the exchange code is the same for both halves (no selects),
//    only table addresses differ; the eleven volume points are taken two at a time the same way;
//  * basis values come from LDS tables indexed by (node permutation of the face, point, h): the
//    neighbour-side basis needs no evaluation;
//  * fluxes in the own tet's frame, accumulators visited once per face, G-form volume term and
//    coalesced row stores as in k_rhs_p2.
__device__ __forceinline__ double pair_swap(double x)
{
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const int l2 = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
  const int h2 = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(h2, l2);
}
// rank of the ordered triple (m0, m1, m2) of distinct local node ids among the 24 possible
__host__ __device__ __forceinline__ int perm_rank(int code)
{
  const int m0 = code & 3, m1 = (code >> 2) & 3, m2 = (code >> 4) & 3;
  const int r1 = m1 - (m1 > m0), r2 = m2 - (m2 > m0) - (m2 > m1);
  return m0 * 6 + r1 * 2 + r2;
}

template <int PROB, int MODE>
__global__ __launch_bounds__(256, 2) void k_rhs_p2s(DevMesh m, Phys ph, double t,
                                                    const double* __restrict__ U,
                                                    double* __restrict__ R,
                                                    double* __restrict__ blockmin,
                                                    double rk_a, double rk_b,
                                                    const double* __restrict__ dtp,
                                                    const double* __restrict__ Un)
{
  constexpr int NDOF = 10, KH = 5, NPROP = NCOMP * NDOF, TPB = 128;
  constexpr bool WITH_DT = MODE == 1, FUSE_RK = MODE == 2;
  __shared__ __attribute__((aligned(16))) P2Split S;
  __shared__ __attribute__((aligned(16))) double stage[TPB * NPROP];
  const int tid = threadIdx.x, h = tid & 1, tl = tid >> 1;
  {
    const double2* src = reinterpret_cast<const double2*>(&g_p2s);
    double2* dst = reinterpret_cast<double2*>(&S);
    for (int i = tid; i < (int)(sizeof(P2Split) / 16); i += 256) dst[i] = src[i];
  }
  const int blk = xcd_tile(blockIdx.x, gridDim.x);
  const int e0 = blk * TPB + tl;
  const bool active = e0 < m.nie;
  const int e = active ? e0 : m.nie - 1;       // every lane runs to the barriers
  const int stride = m.stride;
  double delt = 0.0;

  double acc[NCOMP][KH], u[NCOMP][KH];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c)
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      acc[c][k] = 0.0;
      u[c][k] = U[(size_t)e * NPROP + c * NDOF + KH * h + k];
    }
  __syncthreads();

  const int gb0 = 3 * h, gb1 = 3 - 3 * h;      // first point of this lane's slots 0-2 / 3-5

  // ---- faces ------------------------------------------------------------
  // a face's connectivity is requested one face ahead: neighbour id -> neighbour row would be
  // two dependent memory latencies per face otherwise
  int nbN = m.nbr[e], infoN = m.finfo[e], fN = m.fid[e];
#pragma unroll 1
  for (int lf = 0; lf < 4; ++lf) {
    const int nb = nbN, info = infoN, f = fN;
    if (lf < 3) {
      nbN = m.nbr[(size_t)(lf + 1) * stride + e];
      infoN = m.finfo[(size_t)(lf + 1) * stride + e];
      fN = m.fid[(size_t)(lf + 1) * stride + e];
    }
    if (nb == -1 && !WITH_DT) continue;       // boundary face without a BC (dt still counts it)
    double gq[4];
    load_row<4>(m.fgeo, f, gq);
    const bool own_left = (info >> 6) & 1;
    const double area = gq[0];
    const double osg = own_left ? 1.0 : -1.0;
    const double fn[3] = { osg * gq[1], osg * gq[2], osg * gq[3] };
    const int code_o = lpofa(lf, 0) | (lpofa(lf, 1) << 2) | (lpofa(lf, 2) << 4);
    const double* To = &S.face[perm_rank(code_o)][0][h][0];
    const double* Tn = &S.face[perm_rank(info & 63)][0][h][0];

    double so[3][NCOMP], sn[3][NCOMP];
    if (nb >= 0) {
      double un[NCOMP][KH];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c)
#pragma unroll
        for (int k = 0; k < KH; ++k) un[c][k] = U[(size_t)nb * NPROP + c * NDOF + KH * h + k];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        // slot j (this lane's point) and slot j + 3 (the partner's): partial sums over this
        // lane's modes, the partner's goes across
        const double* Ba = Tn + (gb0 + j) * 12;
        const double* Bb = Tn + (gb1 + j) * 12;
        const double a0 = Ba[0], a1 = Ba[1], a2 = Ba[2], a3 = Ba[3], a4 = Ba[4];
        const double b0 = Bb[0], b1 = Bb[1], b2 = Bb[2], b3 = Bb[3], b4 = Bb[4];
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
          const double pa = un[c][0] * a0 + un[c][1] * a1 + un[c][2] * a2 + un[c][3] * a3 + un[c][4] * a4;
          const double pb = un[c][0] * b0 + un[c][1] * b1 + un[c][2] * b2 + un[c][3] * b3 + un[c][4] * b4;
          sn[j][c] = pa + pair_swap(pb);
        }
      }
    }
    {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        // slot j (this lane's point) and slot j + 3 (the partner's): partial sums over this
        // lane's modes, the partner's goes across
        const double* Ba = To + (gb0 + j) * 12;
        const double* Bb = To + (gb1 + j) * 12;
        const double a0 = Ba[0], a1 = Ba[1], a2 = Ba[2], a3 = Ba[3], a4 = Ba[4];
        const double b0 = Bb[0], b1 = Bb[1], b2 = Bb[2], b3 = Bb[3], b4 = Bb[4];
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
          const double pa = u[c][0] * a0 + u[c][1] * a1 + u[c][2] * a2 + u[c][3] * a3 + u[c][4] * a4;
          const double pb = u[c][0] * b0 + u[c][1] * b1 + u[c][2] * b2 + u[c][3] * b3 + u[c][4] * b4;
          so[j][c] = pa + pair_swap(pb);
        }
      }
    }
    if (nb < 0) {
      ElemGeom g;                             // (not kept across the face loop: 24 registers)
      load_geom(m, e, g);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double* q = S.fq[gb0 + j];
        double P[3];
        face_point(g, lf, q[0], q[1], q[2], P);
        bc_state<PROB>(ph, -nb - 1, so[j], P[0], P[1], P[2], t, fn, sn[j]);
      }
    }
    // fluxes at this lane's three points (own frame, weighted: the own tet loses what leaves)
    const double wsel = (nb == -1) ? 0.0 : 1.0;
    double F[3][NCOMP];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Prim qo, qn;
      primitives(ph, fn, so[j], qo);
      primitives(ph, fn, sn[j], qn);
      const double wq = S.fq[gb0 + j][3] * area;
      if (WITH_DT) {
        // std::max(dSV_l, dSV_r) as (a < b) ? b : a in STORED (left, right) order
        const double d_o = wq * (fabs(qo.vn) + qo.a);
        const double d_n = (nb < 0) ? 0.0 : wq * (fabs(qn.vn) + qn.a);
        const bool take_n = own_left ? (d_o < d_n) : !(d_n < d_o);
        delt += take_n ? d_n : d_o;
      }
      double fl[NCOMP];
      if (ph.flux == 1) flux_lf_q(fn, so[j], sn[j], qo, qn, fl);
      else flux_hllc_own(fn, so[j], sn[j], qo, qn, own_left, fl);
      const double wt = -wq * wsel;
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) F[j][c] = wt * fl[c];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      // this lane's point (slot j) and the partner's (slot j + 3, its flux comes across)
      const double* Ba = To + (gb0 + j) * 12;
      const double* Bb = To + (gb1 + j) * 12;
      const double a0 = Ba[0], a1 = Ba[1], a2 = Ba[2], a3 = Ba[3], a4 = Ba[4];
      const double b0 = Bb[0], b1 = Bb[1], b2 = Bb[2], b3 = Bb[3], b4 = Bb[4];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const double fa = F[j][c], fb = pair_swap(F[j][c]);
        acc[c][0] = fma(fa, a0, fma(fb, b0, acc[c][0]));
        acc[c][1] = fma(fa, a1, fma(fb, b1, acc[c][1]));
        acc[c][2] = fma(fa, a2, fma(fb, b2, acc[c][2]));
        acc[c][3] = fma(fa, a3, fma(fb, b3, acc[c][3]));
        acc[c][4] = fma(fa, a4, fma(fb, b4, acc[c][4]));
      }
    }
  }

  const double vol = m.vol[e];

  // ---- volume (+ source) integral, two points per step: this lane's and its partner's ----
  {
    ElemGeom g;
    load_geom(m, e, g);
    double ji[3][3];
    inverse_jacobian(g, ji);
#pragma unroll 1
    for (int s = 0; s < 6; ++s) {
      const int gm = 2 * s + h, gp = 2 * s + 1 - h;
      const double* tm = S.vol[gm][h];
      const double* tp = S.vol[gp][h];
      double sf[NCOMP];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const double pm = u[c][0] * tm[0] + u[c][1] * tm[1] + u[c][2] * tm[2] + u[c][3] * tm[3] + u[c][4] * tm[4];
        const double pp = u[c][0] * tp[0] + u[c][1] * tp[1] + u[c][2] * tp[2] + u[c][3] * tp[3] + u[c][4] * tp[4];
        sf[c] = pm + pair_swap(pp);
      }
      const double ir = fast_rcp(sf[0]);
      const double uu = sf[1] * ir, vv = sf[2] * ir, ww = sf[3] * ir;
      const double p = eos_pressure(ph, sf[0], uu, vv, ww, sf[4]);
      const double wt = S.vw[gm] * vol;
      const double hh = sf[4] + p;
      // Euler flux F[c][d], src/PDE/CompFlow/DGCompFlow.hpp:599-635
      const double Fv[NCOMP][3] = {
        { sf[1], sf[2], sf[3] },
        { sf[1] * uu + p, sf[2] * uu, sf[3] * uu },
        { sf[1] * vv, sf[2] * vv + p, sf[3] * vv },
        { sf[1] * ww, sf[2] * ww, sf[3] * ww + p },
        { uu * hh, vv * hh, ww * hh } };
      double Gm[NCOMP][3], Gp[NCOMP][3];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          Gm[c][j] = wt * (Fv[c][0] * ji[j][0] + Fv[c][1] * ji[j][1] + Fv[c][2] * ji[j][2]);
          Gp[c][j] = pair_swap(Gm[c][j]);
        }
#pragma unroll
      for (int k = 0; k < KH; ++k) {
        const double m0 = tm[5 + k], m1 = tm[10 + k], m2 = tm[15 + k];
        const double p0 = tp[5 + k], p1 = tp[10 + k], p2 = tp[15 + k];
#pragma unroll
        for (int c = 0; c < NCOMP; ++c)
          acc[c][k] += (Gm[c][0] * m0 + Gm[c][1] * m1 + Gm[c][2] * m2)
                     + (Gp[c][0] * p0 + Gp[c][1] * p1 + Gp[c][2] * p2);
      }
      if constexpr (prob_has_source<PROB>()) {
        // src/PDE/Integrate/Source.cpp:21-141
        const double xi = S.vc[gm][0], eta = S.vc[gm][1], zeta = S.vc[gm][2];
        const double w0 = 1.0 - xi - eta - zeta;
        double P[3], sr[NCOMP];
#pragma unroll
        for (int d = 0; d < 3; ++d)
          P[d] = g.p[0][d] * w0 + g.p[1][d] * xi + g.p[2][d] * eta + g.p[3][d] * zeta;
        prob_src<PROB>(ph, P[0], P[1], P[2], t, sr);
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
          const double wm = wt * sr[c];
          const double wp = pair_swap(wm);
#pragma unroll
          for (int k = 0; k < KH; ++k) acc[c][k] += wm * tm[k] + wp * tp[k];
        }
      }
    }
  }

  if constexpr (FUSE_RK) {
    const double dtv = dtp[0] / vol;
    const double imf[KH] = { h ? 21.0 : 1.0, h ? 14.0 : 10.0, h ? 7.0 : 10.0 / 3.0,
                             h ? 14.0 / 3.0 : 5.0 / 3.0, h ? 7.0 / 3.0 : 35.0 };
#pragma unroll
    for (int c = 0; c < NCOMP; ++c)
#pragma unroll
      for (int k = 0; k < KH; ++k) {
        const double un = Un[(size_t)e * NPROP + c * NDOF + KH * h + k];
        acc[c][k] = rk_a * un + rk_b * (u[c][k] + dtv * imf[k] * acc[c][k]);
      }
  }
  // rows out, coalesced (see k_rhs_p1v)
  {
#pragma unroll
    for (int c = 0; c < NCOMP; ++c)
#pragma unroll
      for (int k = 0; k < KH; ++k) stage[tl * NPROP + c * NDOF + KH * h + k] = acc[c][k];
    __syncthreads();
    const int r0 = blk * TPB;
    const int nrow = (m.nie - r0 < TPB) ? m.nie - r0 : TPB;
    const double2* src = reinterpret_cast<const double2*>(stage);
    double2* dst = reinterpret_cast<double2*>(R + (size_t)r0 * NPROP);
    const int nvalid = nrow * (NPROP / 2);
#pragma unroll
    for (int j = 0; j < (TPB * NPROP / 2 + 255) / 256; ++j) {
      const int i = j * 256 + tid;
      if (i < nvalid) dst[i] = src[i];
    }
  }
  if (WITH_DT) {
    delt += pair_swap(delt);
    double dte = active ? vol / delt : DBL_MAX;
    for (int off = 32; off > 0; off >>= 1) dte = fmin(dte, __shfl_down(dte, off, 64));
    __shared__ double wmin[4];
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) wmin[wv] = dte;
    __syncthreads();
    if (tid == 0)
      blockmin[blockIdx.x] = fmin(fmin(wmin[0], wmin[1]), fmin(wmin[2], wmin[3]));
  }
}

// ------------------------------------------------- DG-P1 RHS (headline kernel)
// Same algorithm as k_rhs<4>, specialised for throughput:
//  * own DOFs live in registers (one coalesced pass), the neighbour's 20 DOFs
//    of face lf+1 are gathered while face lf is computed (software prefetch),
//    so each wave has ~25 independent loads in flight instead of a dependent
//    chain of 60 gathers per face;
//  * 1/x and sqrt from hardware seeds + Newton steps;
//  * dB/dx is constant on a P1 tet: the volume integral accumulates the
//    quadrature-weighted Euler flux once and contracts it with dB/dx at the end;
//  * WITH_DT (RK stage 0): the CFL sum of dg::CompFlow::dt
//    (DGCompFlow.hpp:206-406) is accumulated from the wave speeds the Riemann
//    solver already has -- the separate dt face loop disappears.
template <bool WITH_DT, bool FUSE_RK, int PROB>
__global__ __launch_bounds__(256, QDG_P1_WAVES) void k_rhs_p1(DevMesh m, Phys ph, double t,
                                                const double* __restrict__ U,
                                                double* __restrict__ R,
                                                double* __restrict__ blockmin,
                                                double rk_a, double rk_b,
                                                const double* __restrict__ dtp,
                                                const double* __restrict__ Un)
{
  constexpr int NDOF = 4, NGF = 3, NGV = 5, NPROP = NCOMP * NDOF;
  const Tables<4>& T = c_tab4;
  const int stride = m.stride;
  const int e0 = xcd_tile(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  const bool active = e0 < m.nie;
  const int e = active ? e0 : m.nie - 1;
  double dte = DBL_MAX;
  STAMP_INIT;

  // ---- load schedule -------------------------------------------------------
  // level 1 (independent): own row, the 4 neighbour ids / face codes / face ids,
  //   the 4 node ids, the volume;
  // level 2 (needs level 1): node coordinates, neighbour row + geometry of face 0;
  // then the volume term runs while level 2 for face 0 is still in flight, and
  // inside the face loop the row + geometry of face lf+1 are requested before
  // face lf is computed.  A wave therefore exposes two memory latencies in
  // total instead of two per face.
  double u[NCOMP][NDOF], acc[NCOMP][NDOF];
  load_row<NPROP>(U, e, &u[0][0]);
  const int nb0 = m.nbr[e], nb1 = m.nbr[(size_t)stride + e], nb2 = m.nbr[(size_t)2 * stride + e],
            nb3 = m.nbr[(size_t)3 * stride + e];
  const int in0 = m.finfo[e], in1 = m.finfo[(size_t)stride + e], in2 = m.finfo[(size_t)2 * stride + e],
            in3 = m.finfo[(size_t)3 * stride + e];
  const int f0 = m.fid[e], f1 = m.fid[(size_t)stride + e], f2 = m.fid[(size_t)2 * stride + e],
            f3 = m.fid[(size_t)3 * stride + e];
  const int n0 = m.inpoel[e], n1 = m.inpoel[(size_t)stride + e], n2 = m.inpoel[(size_t)2 * stride + e],
            n3 = m.inpoel[(size_t)3 * stride + e];
  const double vol = m.vol[e];
  STAMP(0);

  double nxt[NCOMP][NDOF], gnx[4];
  load_row<NPROP>(U, nb0 >= 0 ? nb0 : e, &nxt[0][0]);
  load_row<4>(m.fgeo, f0, gnx);
  ElemGeom g;
  {
    double q[4];
    load_row<4>(m.xyz4, n0, q); g.p[0][0] = q[0]; g.p[0][1] = q[1]; g.p[0][2] = q[2];
    load_row<4>(m.xyz4, n1, q); g.p[1][0] = q[0]; g.p[1][1] = q[1]; g.p[1][2] = q[2];
    load_row<4>(m.xyz4, n2, q); g.p[2][0] = q[0]; g.p[2][1] = q[1]; g.p[2][2] = q[2];
    load_row<4>(m.xyz4, n3, q); g.p[3][0] = q[0]; g.p[3][1] = q[1]; g.p[3][2] = q[2];
  }
  STAMP(1);

#pragma unroll
  for (int c = 0; c < NCOMP; ++c)
#pragma unroll
    for (int k = 0; k < NDOF; ++k) acc[c][k] = 0.0;

  // ---- volume integral: dB/dx constant on a P1 tet --------------------------
  {
    double ji[3][3];
    inverse_jacobian(g, ji);
    double Fs[NCOMP][3];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) Fs[c][0] = Fs[c][1] = Fs[c][2] = 0.0;
#pragma unroll
    for (int ig = 0; ig < NGV; ++ig) {
      double s[NCOMP];
      state_from<NDOF>(u, T.vB[ig], s);
      const double ir = fast_rcp(s[0]);
      const double uu = s[1] * ir, vv = s[2] * ir, ww = s[3] * ir;
      const double p = eos_pressure(ph, s[0], uu, vv, ww, s[4]);
      const double wg = T.vw[ig];
      const double h = s[4] + p;
      Fs[0][0] += wg * s[1];            Fs[0][1] += wg * s[2];            Fs[0][2] += wg * s[3];
      Fs[1][0] += wg * (s[1] * uu + p); Fs[1][1] += wg * (s[2] * uu);     Fs[1][2] += wg * (s[3] * uu);
      Fs[2][0] += wg * (s[1] * vv);     Fs[2][1] += wg * (s[2] * vv + p); Fs[2][2] += wg * (s[3] * vv);
      Fs[3][0] += wg * (s[1] * ww);     Fs[3][1] += wg * (s[2] * ww);     Fs[3][2] += wg * (s[3] * ww + p);
      Fs[4][0] += wg * (uu * h);        Fs[4][1] += wg * (vv * h);        Fs[4][2] += wg * (ww * h);
    }
#pragma unroll
    for (int k = 1; k < NDOF; ++k) {
      const double g0 = T.vdB[0][0][k], g1 = T.vdB[0][1][k], g2 = T.vdB[0][2][k];
      const double dx = vol * (g0 * ji[0][0] + g1 * ji[1][0] + g2 * ji[2][0]);
      const double dy = vol * (g0 * ji[0][1] + g1 * ji[1][1] + g2 * ji[2][1]);
      const double dz = vol * (g0 * ji[0][2] + g1 * ji[1][2] + g2 * ji[2][2]);
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) acc[c][k] += Fs[c][0] * dx + Fs[c][1] * dy + Fs[c][2] * dz;
    }
  }

  // ---- source integral (manufactured-solution problems only) ----------------
  if constexpr (prob_has_source<PROB>()) {
#pragma unroll 1
    for (int ig = 0; ig < NGV; ++ig) {
      const double xi = T.vc[ig][0], eta = T.vc[ig][1], zeta = T.vc[ig][2];
      const double w0 = 1.0 - xi - eta - zeta;
      double P[3], s[NCOMP];
#pragma unroll
      for (int d = 0; d < 3; ++d)
        P[d] = g.p[0][d] * w0 + g.p[1][d] * xi + g.p[2][d] * eta + g.p[3][d] * zeta;
      prob_src<PROB>(ph, P[0], P[1], P[2], t, s);
      const double wt = T.vw[ig] * vol;
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const double ws = wt * s[c];
        acc[c][0] += ws;
#pragma unroll
        for (int k = 1; k < NDOF; ++k) acc[c][k] += ws * T.vB[ig][k];
      }
    }
  }
  STAMP(2);

  // ---- faces ----------------------------------------------------------------
  double delt = 0.0;
  constexpr bool HAS_DIRICHLET = (PROB == 3 || PROB == 4 || PROB == 0 || PROB == 7 || PROB == 10);
#pragma unroll 1
  for (int lf = 0; lf < 4; ++lf) {
    const int nb = (lf == 0) ? nb0 : (lf == 1) ? nb1 : (lf == 2) ? nb2 : nb3;
    const int info = (lf == 0) ? in0 : (lf == 1) ? in1 : (lf == 2) ? in2 : in3;
    double cur[NCOMP][NDOF];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c)
#pragma unroll
      for (int k = 0; k < NDOF; ++k) cur[c][k] = nxt[c][k];
    const double area = gnx[0];
    const double fn[3] = { gnx[1], gnx[2], gnx[3] };
    if (lf < 3) {
      const int nbn = (lf == 0) ? nb1 : (lf == 1) ? nb2 : nb3;
      const int fnx_ = (lf == 0) ? f1 : (lf == 1) ? f2 : f3;
      load_row<NPROP>(U, nbn >= 0 ? nbn : e, &nxt[0][0]);
      load_row<4>(m.fgeo, fnx_, gnx);
    }
    STAMP(3);
    const bool own_left = (info >> 6) & 1;
    // Boundary faces run through the SAME straight-line code as interior ones
    // (no wave divergence): the "neighbour" state is the own state, mirrored
    // for Symmetry (DGCompFlow.hpp:672-690: u_r = u_l - 2 (u_l.n) n, same rho
    // and rhoE), untouched for Extrapolate; a face without a configured BC
    // gets weight 0.  Only Dirichlet needs the analytic solution (a real
    // branch, compiled in for the manufactured-solution problems only).
    const bool bnd = nb < 0;
    const int bc = bnd ? -nb - 1 : 0;
    const double refl = (bc == 2) ? 2.0 : 0.0;
    const double wsel = (bnd && bc == 0) ? 0.0 : 1.0;
#pragma unroll 1
    for (int ig = 0; ig < NGF; ++ig) {
      const double s0 = T.fs[ig][0], s1 = T.fs[ig][1], s2 = T.fs[ig][2];
      double so[NCOMP], sn[NCOMP], fl[NCOMP];
      state_from<NDOF>(u, T.fB[lf][ig], so);
      {
        double xi, eta, zeta, Bn[NDOF];
        nbr_ref_coords(info, s0, s1, s2, xi, eta, zeta);
        eval_basis<NDOF>(xi, eta, zeta, Bn);
        state_from<NDOF>(cur, Bn, sn);
      }
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) sn[c] = bnd ? so[c] : sn[c];
      {
        const double vn2 = refl * (sn[1] * fn[0] + sn[2] * fn[1] + sn[3] * fn[2]);
        sn[1] -= vn2 * fn[0]; sn[2] -= vn2 * fn[1]; sn[3] -= vn2 * fn[2];
      }
      if constexpr (HAS_DIRICHLET) {
        if (bc == 1) {
          double P[3];
          face_point(g, lf, s0, s1, s2, P);
          prob_solution<PROB>(ph, P[0], P[1], P[2], t, sn);
        }
      }
      double L[NCOMP], Rr[NCOMP];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) { L[c] = own_left ? so[c] : sn[c]; Rr[c] = own_left ? sn[c] : so[c]; }
      Prim ql, qr;
      primitives(ph, fn, L, ql);
      primitives(ph, fn, Rr, qr);
      const double wq = T.fw[ig] * area;
      if (WITH_DT) {
        // delt += std::max(dSV_l, dSV_r); boundary faces: dSV_r = 0
        const double dl = wq * (fabs(ql.vn) + ql.a);
        const double dr = bnd ? 0.0 : wq * (fabs(qr.vn) + qr.a);
        delt += (dl < dr) ? dr : dl;
      }
      if (ph.flux == 1) flux_lf_q(fn, L, Rr, ql, qr, fl);
      else flux_hllc_q(fn, L, Rr, ql, qr, fl);
      const double wt = (own_left ? -wq : wq) * wsel;
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const double wf = wt * fl[c];
        acc[c][0] += wf;
#pragma unroll
        for (int k = 1; k < NDOF; ++k) acc[c][k] += wf * T.fB[lf][ig][k];
      }
    }
    STAMP(4);
  }

  if (FUSE_RK) {
    // SSP-RK3 stage update fused into the RHS (stages 1 and 2, dt known):
    // `R` is the NEW state buffer, R itself never goes to memory
    //   U_new = a*Un + b*(U + dt*R/L),  L = vol*massfac[k]   (DG.cpp:1478-1488)
    constexpr double imf[4] = { 1.0, 10.0, 10.0 / 3.0, 5.0 / 3.0 };
    const double dtv = dtp[0] / vol;
    double un[NCOMP][NDOF];
    load_row<NPROP>(Un, e, &un[0][0]);
#pragma unroll
    for (int c = 0; c < NCOMP; ++c)
#pragma unroll
      for (int k = 0; k < NDOF; ++k)
        acc[c][k] = rk_a * un[c][k] + rk_b * (u[c][k] + dtv * imf[k] * acc[c][k]);
  }
  if (active) store_row<NPROP>(R, e, &acc[0][0]);
  STAMP(5);

  if (WITH_DT) {
    if (active) dte = vol / delt;
    for (int off = 32; off > 0; off >>= 1) dte = fmin(dte, __shfl_down(dte, off, 64));
    __shared__ double wmin[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wmin[wv] = dte;
    __syncthreads();
    if (threadIdx.x == 0)
      blockmin[blockIdx.x] = fmin(fmin(wmin[0], wmin[1]), fmin(wmin[2], wmin[3]));
  }
}

// ------------------------------------------- DG-P1 RHS, tile / face-task form
// Every face of a 248-tet tile is evaluated ONCE: a face whose two tets lie in
// the tile (about three quarters of all interior faces of a Morton-ordered
// tile) is computed by one lane, which adds the flux integral to BOTH tets'
// accumulators in LDS; faces towards other tiles, ghosts or the physical
// boundary are computed by their in-tile tet as before.  Compared with the
// element-centric kernel (k_rhs_p1) a tile evaluates ~38 % fewer Riemann
// problems.  Work items are dense (host-built task lists, sorted by kind and
// local face), so the saving is real SIMD time, not idle lanes.
//
// P1 algebra used to keep the per-task state small (60 instead of 160 VGPRs of
// persistent data): with s_j(g) the barycentric weights of Gauss point g on the
// face's three vertices,
//    state(g)        = sum_j s_j(g) * V_j          V_j = state at face vertex j
//    R_i[c][k]      -+= sum_j W_j[c] * Bv_i,j[k]    W_j[c] = sum_g s_j(g) w_g A F_c(g)
// where Bv_i,j is tet i's basis at face vertex j: one W serves both tets.
//
// The LDS accumulation uses ds_add_f64: the order in which the (at most four)
// face contributions of a tet arrive is not fixed, so R can differ in the last
// bit from run to run; QDG_DETERMINISTIC_RHS=1 selects k_rhs_p1 instead.
// LDS index of (tet, vertex, component) in the tile kernels' nodal arrays
#define LIDX(e, v, c) ((((v) * NCOMP) + (c)) * TILE + (e))

__device__ __forceinline__ void vertex_basis(int v, double& b1, double& b2, double& b3)
{
  // B1 = 2xi+eta+zeta-1, B2 = 3eta+zeta-1, B3 = 4zeta-1 at reference vertex v
  b1 = (v == 0) ? -1.0 : (v == 1) ? 1.0 : 0.0;
  b2 = (v == 2) ? 2.0 : (v == 3) ? 0.0 : -1.0;
  b3 = (v == 3) ? 3.0 : -1.0;
}

// PDG (p-adaptive DG, scheme pdg): a tet with m.ndofel == 1 is a P0 element --
// its state is its mean (Surface.cpp:146-156), only its mean is updated
// (update_rhs_fa, Surface.cpp:234-271), it has no volume term (Volume.cpp:56)
// and its source integral uses the 1-point rule (Source.cpp:54).  The face
// quadrature keeps 3 points where the reference takes max(ng_l, ng_r)
// (Surface.cpp:81-86): between two P0 tets both states are constant, so the 1-
// and the 3-point sums agree to rounding.
template <bool WITH_DT, bool FUSE_RK, int PROB, bool PDG>
__global__ __launch_bounds__(TILE_BS, QDG_TILE_WAVES) void k_rhs_p1t(DevMesh m, Phys ph, double t,
                                                     const double* __restrict__ U,
                                                     double* __restrict__ R,
                                                     double* __restrict__ blockmin,
                                                     double rk_a, double rk_b,
                                                     const double* __restrict__ dtp,
                                                     const double* __restrict__ Un)
{
  constexpr int NDOF = 4, NGF = 3, NGV = 5, NPROP = NCOMP * NDOF;
  const Tables<4>& T = c_tab4;
  // LDS: the tile's states in NODAL form, nod[e][vertex][c] (a P1 state is
  // affine: its value at a face point is the barycentric mix of its vertex
  // values), and per-vertex flux accumulators accN[e][vertex][c]
  __shared__ __attribute__((aligned(16))) double nod[TILE * NPROP];
  __shared__ double accN[TILE * NPROP];
  __shared__ double sdelt[WITH_DT ? TILE : 1];
  const int tid = threadIdx.x;
  const int tile = m.blk0 + xcd_tile(blockIdx.x, gridDim.x);
  // fixed TILE-row tiles (the default): no load in front of the tile's own rows
  const int tile_e0 = m.tile_rows ? tile * m.tile_rows : m.tile_row[tile];
  const int nloc = m.tile_rows ? ((m.nie - tile_e0 < m.tile_rows) ? m.nie - tile_e0 : m.tile_rows)
                               : m.tile_row[tile + 1] - tile_e0;

  // this lane's task descriptors (up to MAXT rounds) and the first task's face
  // geometry / external row are requested before anything waits on LDS
  constexpr int MAXT = 4;
  const int t0 = m.tile_off[tile], t1 = m.tile_off[tile + 1];
  int ta[MAXT], tf[MAXT], tn[MAXT];
#pragma unroll
  for (int q = 0; q < MAXT; ++q) {
    // compact lists: the tile's tasks are [t0, t1); padded lists (QDG_TILE_V1=1 on a mesh built for
    // version 2): its slots start at tile * stride, unused ones hold -1
    const size_t it = m.task_stride > 0 ? (size_t)tile * m.task_stride + tid + TILE_BS * q
                                        : (size_t)(t0 + tid + TILE_BS * q);
    const bool ok = m.task_stride > 0 || (int)it < t1;
    ta[q] = ok ? m.task_a[it] : -1;
    tf[q] = ok ? m.task_f[it] : 0;
    tn[q] = ok ? m.task_nb[it] : 0;
  }

  // ---- phase 0: modal row -> the 4 vertex states, accumulators = 0 ------------
  if (tid < TILE) {
    double r[NCOMP][NDOF];
    if (tid < nloc) load_row<NPROP>(U, tile_e0 + tid, &r[0][0]);
    else {
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) { r[c][0] = 1.0; r[c][1] = r[c][2] = r[c][3] = 0.0; }
    }
    if constexpr (PDG) {
      if (tid < nloc && m.ndofel[tile_e0 + tid] == 1) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) r[c][1] = r[c][2] = r[c][3] = 0.0;
      }
    }
    double v[4][NCOMP];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      // B at the vertices: v0 (-1,-1,-1), v1 (1,-1,-1), v2 (0,2,-1), v3 (0,0,3)
      const double a = r[c][0] - r[c][3];
      v[0][c] = a - r[c][1] - r[c][2];
      v[1][c] = a + r[c][1] - r[c][2];
      v[2][c] = a + 2.0 * r[c][2];
      v[3][c] = r[c][0] + 3.0 * r[c][3];
    }
    // LDS planes [vertex][component][tet]: lanes of a wave work on different tets at the
    // same (vertex, component), so tet-fastest storage is free of bank conflicts
#pragma unroll
    for (int vx = 0; vx < 4; ++vx)
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) { nod[LIDX(tid, vx, c)] = v[vx][c]; accN[LIDX(tid, vx, c)] = 0.0; }
    if (WITH_DT) sdelt[tid] = 0.0;
  }
  double gnx[4], rnx[NCOMP][NDOF];
  int ndnx = 4;                      // PDG: ndofel of the external neighbour
  if (ta[0] >= 0) {
    load_row<4>(m.fgeo, tf[0], gnx);
    if (TASK_KIND(ta[0]) == TASK_EXT) {
      load_row<NPROP>(U, tn[0], &rnx[0][0]);
      if constexpr (PDG) ndnx = m.ndofel[tn[0]];
    }
  }
  __syncthreads();

  // ---- phase 1: one lane per face task ------------------------------------------
  constexpr bool HAS_DIRICHLET = (PROB == 3 || PROB == 4 || PROB == 0 || PROB == 7 || PROB == 10);
  if ((t1 - t0) > TILE_BS * MAXT) __builtin_trap();   // cannot happen: <= 4*TILE tasks per tile
#pragma unroll 1
  for (int q = 0; q < MAXT; ++q) {
    const int a = (q == 0) ? ta[0] : (q == 1) ? ta[1] : (q == 2) ? ta[2] : ta[3];
    if (a < 0) break;
    const int el = TASK_EL(a), lf = TASK_LF(a), code = TASK_CODE(a), kind = TASK_KIND(a),
              bc = TASK_BC(a), pl = TASK_PL(a);
    const bool own_left = TASK_OWNLEFT(a);
    const double area = gnx[0];
    const double fn[3] = { gnx[1], gnx[2], gnx[3] };
    double rex[NCOMP][NDOF];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c)
#pragma unroll
      for (int k = 0; k < NDOF; ++k) rex[c][k] = rnx[c][k];
    if constexpr (PDG) {
      if (ndnx == 1) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) rex[c][1] = rex[c][2] = rex[c][3] = 0.0;
      }
    }
    {
      // prefetch the next task of this lane
      const int an = (q == 0) ? ta[1] : (q == 1) ? ta[2] : (q == 2) ? ta[3] : -1;
      const int fq = (q == 0) ? tf[1] : (q == 1) ? tf[2] : tf[3];
      const int nq = (q == 0) ? tn[1] : (q == 1) ? tn[2] : tn[3];
      if (an >= 0) {
        load_row<4>(m.fgeo, fq, gnx);
        if (TASK_KIND(an) == TASK_EXT) {
          load_row<NPROP>(U, nq, &rnx[0][0]);
          if constexpr (PDG) ndnx = m.ndofel[nq];
        }
      }
    }
    const bool bnd = kind == TASK_BND;

    // vertex states of both tets at the three face vertices
    double Vo[3][NCOMP], Vn[3][NCOMP];
    int no[3], nn[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { no[j] = lpofa(lf, j); nn[j] = (code >> (2 * j)) & 3; }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) Vo[j][c] = nod[LIDX(el, no[j], c)];
    }
    if (kind == TASK_INT) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) Vn[j][c] = nod[LIDX(pl, nn[j], c)];
      }
    } else if (kind == TASK_EXT) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double b1, b2, b3;
        vertex_basis(nn[j], b1, b2, b3);
#pragma unroll
        for (int c = 0; c < NCOMP; ++c)
          Vn[j][c] = rex[c][0] + rex[c][1] * b1 + rex[c][2] * b2 + rex[c][3] * b3;
      }
    } else {
      // Extrapolate: u_r = u_l; Symmetry: mirrored momentum (DGCompFlow.hpp:672-690),
      // a linear map, applied to the vertex states
      const double refl = (bc == 2) ? 2.0 : 0.0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double vn2 = refl * (Vo[j][1] * fn[0] + Vo[j][2] * fn[1] + Vo[j][3] * fn[2]);
        Vn[j][0] = Vo[j][0];
        Vn[j][1] = Vo[j][1] - vn2 * fn[0];
        Vn[j][2] = Vo[j][2] - vn2 * fn[1];
        Vn[j][3] = Vo[j][3] - vn2 * fn[2];
        Vn[j][4] = Vo[j][4];
      }
    }
    const double wsel = (bnd && bc == 0) ? 0.0 : 1.0;   // boundary face without a BC: no flux

    double W[3][NCOMP], dsum = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) W[j][c] = 0.0;

    // PDG: a Dirichlet face of a P0 tet takes NGfa(1) = 1 point (Boundary.cpp:94) --
    // the analytic state varies along the face, so the count must match
    bool one = false;
    if constexpr (PDG && HAS_DIRICHLET) one = bnd && bc == 1 && m.ndofel[tile_e0 + el] == 1;
    const int ngl = one ? 1 : NGF;
#pragma unroll QDG_TILE_GP_UNROLL
    for (int ig = 0; ig < ngl; ++ig) {
      const double s0 = one ? 1.0 / 3.0 : T.fs[ig][0], s1 = one ? 1.0 / 3.0 : T.fs[ig][1],
                   s2 = one ? 1.0 / 3.0 : T.fs[ig][2];
      double so[NCOMP], sn[NCOMP], fl[NCOMP];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        so[c] = s0 * Vo[0][c] + s1 * Vo[1][c] + s2 * Vo[2][c];
        sn[c] = s0 * Vn[0][c] + s1 * Vn[1][c] + s2 * Vn[2][c];
      }
      if constexpr (HAS_DIRICHLET) {
        if (bnd && bc == 1) {
          ElemGeom g;
          load_geom(m, tile_e0 + el, g);
          double P[3];
          face_point(g, lf, s0, s1, s2, P);
          prob_solution<PROB>(ph, P[0], P[1], P[2], t, sn);
        }
      }
      double L[NCOMP], Rr[NCOMP];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) { L[c] = own_left ? so[c] : sn[c]; Rr[c] = own_left ? sn[c] : so[c]; }
      Prim ql, qr;
      primitives(ph, fn, L, ql);
      primitives(ph, fn, Rr, qr);
      const double wq = (one ? 1.0 : T.fw[ig]) * area;
      if (WITH_DT) {
        const double dl = wq * (fabs(ql.vn) + ql.a);
        const double dr = bnd ? 0.0 : wq * (fabs(qr.vn) + qr.a);
        dsum += (dl < dr) ? dr : dl;
      }
      if (ph.flux == 1) flux_lf_q(fn, L, Rr, ql, qr, fl);
      else flux_hllc_q(fn, L, Rr, ql, qr, fl);
      const double wt = wq * wsel;
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const double wf = wt * fl[c];
        W[0][c] += s0 * wf; W[1][c] += s1 * wf; W[2][c] += s2 * wf;
      }
    }

    // ---- scatter the vertex-weighted flux sums: left tet -=, right tet += -------
    {
      const double sg = own_left ? -1.0 : 1.0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c)
          __hip_atomic_fetch_add(accN + LIDX(el, no[j], c), sg * W[j][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (WITH_DT) __hip_atomic_fetch_add(sdelt + el, dsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (kind == TASK_INT) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
          for (int c = 0; c < NCOMP; ++c)
            __hip_atomic_fetch_add(accN + LIDX(pl, nn[j], c), -sg * W[j][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (WITH_DT) __hip_atomic_fetch_add(sdelt + pl, dsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  // phase-2 inputs are requested before the barrier (their latency overlaps the
  // other waves' last tasks)
  double u[NCOMP][NDOF], un[NCOMP][NDOF];
  double vol = 1.0;
  bool p0 = false;                   // PDG: this tet is a P0 element
  ElemGeom g;
  if (tid < nloc) {
    const int e = tile_e0 + tid;
    const int stride = m.stride;
    load_row<NPROP>(U, e, &u[0][0]);          // modal row again (L1/L2 hit)
    if (FUSE_RK) load_row<NPROP>(Un, e, &un[0][0]);
    if constexpr (PDG) p0 = m.ndofel[e] == 1;
    vol = m.vol[e];
    const int n0 = m.inpoel[e], n1 = m.inpoel[(size_t)stride + e], n2 = m.inpoel[(size_t)2 * stride + e],
              n3 = m.inpoel[(size_t)3 * stride + e];
    double q[4];
    load_row<4>(m.xyz4, n0, q); g.p[0][0] = q[0]; g.p[0][1] = q[1]; g.p[0][2] = q[2];
    load_row<4>(m.xyz4, n1, q); g.p[1][0] = q[0]; g.p[1][1] = q[1]; g.p[1][2] = q[2];
    load_row<4>(m.xyz4, n2, q); g.p[2][0] = q[0]; g.p[2][1] = q[1]; g.p[2][2] = q[2];
    load_row<4>(m.xyz4, n3, q); g.p[3][0] = q[0]; g.p[3][1] = q[1]; g.p[3][2] = q[2];
  }
  __syncthreads();

  // ---- phase 2: one lane per tet: volume (+source) term, epilogue, store --------
  double dte = DBL_MAX;
  double acc[NCOMP][NDOF];
  if (tid < nloc) {
    const int e = tile_e0 + tid;
    {
      // R[c][k] = sum_v accN[v][c] * B_k(vertex v)
      double nv[4][NCOMP];
#pragma unroll
      for (int vx = 0; vx < 4; ++vx)
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) nv[vx][c] = accN[LIDX(tid, vx, c)];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        acc[c][0] = (nv[0][c] + nv[1][c]) + (nv[2][c] + nv[3][c]);
        acc[c][1] = nv[1][c] - nv[0][c];
        acc[c][2] = 2.0 * nv[2][c] - nv[0][c] - nv[1][c];
        acc[c][3] = 3.0 * nv[3][c] - nv[0][c] - nv[1][c] - nv[2][c];
      }
    }
    if constexpr (PDG) {
      if (p0) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) u[c][1] = u[c][2] = u[c][3] = 0.0;
      }
    }
    {
      double ji[3][3];
      inverse_jacobian(g, ji);
      double Fs[NCOMP][3];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) Fs[c][0] = Fs[c][1] = Fs[c][2] = 0.0;
#pragma unroll
      for (int ig = 0; ig < NGV; ++ig) {
        double s[NCOMP];
        state_from<NDOF>(u, T.vB[ig], s);
        const double ir = fast_rcp(s[0]);
        const double uu = s[1] * ir, vv = s[2] * ir, ww = s[3] * ir;
        const double p = eos_pressure(ph, s[0], uu, vv, ww, s[4]);
        const double wg = T.vw[ig];
        const double h = s[4] + p;
        Fs[0][0] += wg * s[1];            Fs[0][1] += wg * s[2];            Fs[0][2] += wg * s[3];
        Fs[1][0] += wg * (s[1] * uu + p); Fs[1][1] += wg * (s[2] * uu);     Fs[1][2] += wg * (s[3] * uu);
        Fs[2][0] += wg * (s[1] * vv);     Fs[2][1] += wg * (s[2] * vv + p); Fs[2][2] += wg * (s[3] * vv);
        Fs[3][0] += wg * (s[1] * ww);     Fs[3][1] += wg * (s[2] * ww);     Fs[3][2] += wg * (s[3] * ww + p);
        Fs[4][0] += wg * (uu * h);        Fs[4][1] += wg * (vv * h);        Fs[4][2] += wg * (ww * h);
      }
#pragma unroll
      for (int k = 1; k < NDOF; ++k) {
        const double g0 = T.vdB[0][0][k], g1 = T.vdB[0][1][k], g2 = T.vdB[0][2][k];
        const double dx = vol * (g0 * ji[0][0] + g1 * ji[1][0] + g2 * ji[2][0]);
        const double dy = vol * (g0 * ji[0][1] + g1 * ji[1][1] + g2 * ji[2][1]);
        const double dz = vol * (g0 * ji[0][2] + g1 * ji[1][2] + g2 * ji[2][2]);
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) acc[c][k] += Fs[c][0] * dx + Fs[c][1] * dy + Fs[c][2] * dz;
      }
    }
    if constexpr (PDG) {
      if (p0) {                        // no high-order update of a P0 element
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) acc[c][1] = acc[c][2] = acc[c][3] = 0.0;
      }
    }
    if constexpr (prob_has_source<PROB>()) {
      const int ngs = (PDG && p0) ? 1 : NGV;     // Source.cpp:54: NGvol(ndofel[e])
#pragma unroll 1
      for (int ig = 0; ig < ngs; ++ig) {
        const bool one = PDG && p0;
        const double xi = one ? 0.25 : T.vc[ig][0], eta = one ? 0.25 : T.vc[ig][1],
                     zeta = one ? 0.25 : T.vc[ig][2];
        const double w0 = 1.0 - xi - eta - zeta;
        double P[3], s[NCOMP];
#pragma unroll
        for (int d = 0; d < 3; ++d)
          P[d] = g.p[0][d] * w0 + g.p[1][d] * xi + g.p[2][d] * eta + g.p[3][d] * zeta;
        prob_src<PROB>(ph, P[0], P[1], P[2], t, s);
        const double wt = (one ? 1.0 : T.vw[ig]) * vol;
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
          const double ws = wt * s[c];
          acc[c][0] += ws;
          if (!one) {
#pragma unroll
            for (int k = 1; k < NDOF; ++k) acc[c][k] += ws * T.vB[ig][k];
          }
        }
      }
    }
    if (FUSE_RK) {
      constexpr double imf[4] = { 1.0, 10.0, 10.0 / 3.0, 5.0 / 3.0 };
      const double dtv = dtp[0] / vol;
#pragma unroll
      for (int c = 0; c < NCOMP; ++c)
#pragma unroll
        for (int k = 0; k < NDOF; ++k)
          acc[c][k] = rk_a * un[c][k] + rk_b * (u[c][k] + dtv * imf[k] * acc[c][k]);
    }
    if (WITH_DT) dte = vol / sdelt[tid];
  }
  // rows out through LDS as coalesced wave stores (see k_rhs_p1v)
  __syncthreads();
  if (tid < nloc) {
    double2* row = reinterpret_cast<double2*>(nod + (size_t)tid * NPROP);
#pragma unroll
    for (int j = 0; j < NPROP / 2; ++j) row[j] = make_double2((&acc[0][0])[2 * j], (&acc[0][0])[2 * j + 1]);
  }
  __syncthreads();
  {
    const double2* src = reinterpret_cast<const double2*>(nod);
    double2* dst = reinterpret_cast<double2*>(R + (size_t)tile_e0 * NPROP);
    const int nvalid = nloc * (NPROP / 2);
#pragma unroll
    for (int j = 0; j < NPROP / 2; ++j) {
      const int i = j * TILE_BS + tid;
      if (i < nvalid) dst[i] = src[i];
    }
  }

  if (WITH_DT) {
    for (int off = 32; off > 0; off >>= 1) dte = fmin(dte, __shfl_down(dte, off, 64));
    __shared__ double wmin[TILE_BS / 64];
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) wmin[wv] = dte;
    __syncthreads();
    if (tid == 0) {
      double mn = wmin[0];
      for (int w = 1; w < TILE_BS / 64; ++w) mn = fmin(mn, wmin[w]);
      blockmin[tile] = mn;
    }
  }
}

// ------------------------------------------- DG-P1 RHS, tile / face-task form, version 2
// Same tiles, task lists, LDS layout and phases as k_rhs_p1t; the face task is leaner:
//  * own-frame evaluation with the mirrored HLLC ladder instead of swapping the two states
//    into stored (left, right) order at every Gauss point (20 selects per point);
//  * the 3-point rule's structure (one heavy vertex per point, equal weights): one FMA per
//    state component per point, and the vertex-weighted flux sums formed once after the
//    point loop from the three raw fluxes;
//  * uniform order only (p-adaptive runs use k_rhs_p1t).
template <bool WITH_DT, bool FUSE_RK, int PROB>
__global__ __launch_bounds__(TILE_BS, QDG_TILE_WAVES) void k_rhs_p1v(DevMesh m, Phys ph, double t,
                                                     const double* __restrict__ U,
                                                     double* __restrict__ R,
                                                     double* __restrict__ blockmin,
                                                     double rk_a, double rk_b,
                                                     const double* __restrict__ dtp,
                                                     const double* __restrict__ Un)
{
  constexpr int NDOF = 4, NGF = 3, NGV = 5, NPROP = NCOMP * NDOF;
  const Tables<4>& T = c_tab4;
  // LDS: the tile's states in NODAL form, nod[e][vertex][c] (a P1 state is
  // affine: its value at a face point is the barycentric mix of its vertex
  // values), and per-vertex flux accumulators accN[e][vertex][c]
  __shared__ __attribute__((aligned(16))) double nod[TILE * NPROP];
  __shared__ double accN[TILE * NPROP];
  __shared__ double sdelt[WITH_DT ? TILE : 1];
  const int tid = threadIdx.x;
  const int tile = m.blk0 + xcd_tile(blockIdx.x, gridDim.x);
  const int tile_e0 = m.tile_rows ? tile * m.tile_rows : m.tile_row[tile];
  const int nloc = m.tile_rows ? ((m.nie - tile_e0 < m.tile_rows) ? m.nie - tile_e0 : m.tile_rows)
                               : m.tile_row[tile + 1] - tile_e0;

  // this lane's task descriptors (up to MAXT rounds) and the first task's face
  // geometry / external row are requested before anything waits on LDS
  constexpr int MAXT = 4;
  int ta[MAXT], tf[MAXT], tn[MAXT];
  int t0 = 0, t1 = 0;
  if (m.task_stride > 0) {
    // padded task lists (unused slots hold -1): a tile's slots start at tile * stride, so the
    // descriptors need no offset load in front of them (one dependent memory latency less on
    // the way to the first neighbour row)
    const size_t base = (size_t)tile * m.task_stride;
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
      const size_t it = base + tid + TILE_BS * q;
      ta[q] = m.task_a[it];
      // with face records in task order (tgeo) the slot itself addresses the record
      tf[q] = m.tgeo ? (int)it : m.task_f[it];
      tn[q] = m.task_nb[it];
    }
  } else {
    t0 = m.tile_off[tile]; t1 = m.tile_off[tile + 1];
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
      const int it = t0 + tid + TILE_BS * q;
      const bool ok = it < t1;
      ta[q] = ok ? m.task_a[it] : -1;
      tf[q] = ok ? m.task_f[it] : 0;
      tn[q] = ok ? m.task_nb[it] : 0;
    }
  }

  // ---- phase 0: modal row -> the 4 vertex states, accumulators = 0 ------------
  if (tid < TILE) {
    double r[NCOMP][NDOF];
    if (tid < nloc) load_row<NPROP>(U, tile_e0 + tid, &r[0][0]);
    else {
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) { r[c][0] = 1.0; r[c][1] = r[c][2] = r[c][3] = 0.0; }
    }
    double v[4][NCOMP];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      // B at the vertices: v0 (-1,-1,-1), v1 (1,-1,-1), v2 (0,2,-1), v3 (0,0,3)
      const double a = r[c][0] - r[c][3];
      v[0][c] = a - r[c][1] - r[c][2];
      v[1][c] = a + r[c][1] - r[c][2];
      v[2][c] = a + 2.0 * r[c][2];
      v[3][c] = r[c][0] + 3.0 * r[c][3];
    }
    // LDS planes [vertex][component][tet]: lanes of a wave work on different tets at the
    // same (vertex, component), so tet-fastest storage is free of bank conflicts
#pragma unroll
    for (int vx = 0; vx < 4; ++vx)
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) { nod[LIDX(tid, vx, c)] = v[vx][c]; accN[LIDX(tid, vx, c)] = 0.0; }
    if (WITH_DT) sdelt[tid] = 0.0;
  }
  double gnx[4], rnx[NCOMP][NDOF];
  const double* __restrict__ geo = (m.task_stride > 0 && m.tgeo) ? m.tgeo : m.fgeo;
  if (ta[0] >= 0) {
    load_row<4>(geo, tf[0], gnx);
    if (TASK_KIND(ta[0]) == TASK_EXT) load_row<NPROP>(U, tn[0], &rnx[0][0]);
  }
  __syncthreads();

  // ---- phase 1: one lane per face task ------------------------------------------
  constexpr bool HAS_DIRICHLET = (PROB == 3 || PROB == 4 || PROB == 0 || PROB == 7 || PROB == 10);
  if ((t1 - t0) > TILE_BS * MAXT) __builtin_trap();   // cannot happen: <= 4*TILE tasks per tile
#ifdef QDG_KO_ATOM
  double ko_sum = 0.0;
#endif
#pragma unroll 1
  for (int q = 0; q < MAXT; ++q) {
#ifdef QDG_KO_TASKS
    break;
#endif
    const int a = (q == 0) ? ta[0] : (q == 1) ? ta[1] : (q == 2) ? ta[2] : ta[3];
    if (a < 0) break;
    const int el = TASK_EL(a), lf = TASK_LF(a), code = TASK_CODE(a), kind = TASK_KIND(a),
              bc = TASK_BC(a), pl = TASK_PL(a);
    const bool own_left = TASK_OWNLEFT(a);
    // everything below works in the OWN tet's frame: left' = own, right' = neighbour,
    // n' = the own tet's outward normal (the stored normal or its negative).  For a face
    // whose stored left tet is the neighbour this is the mirror image of the reference's
    // evaluation: wave speeds change sign (Sl' = -Sr, Sm' = -Sm, Sr' = -Sl), so the
    // reference's ladder (HLLC.hpp:93-124) is applied in its mirrored form -- same four
    // fluxes, same fall-through of a NaN wave speed to the STORED right state -- and the
    // own tet always loses what the neighbour gains.
    const double area = gnx[0];
    const double osg = own_left ? 1.0 : -1.0;
    const double fn[3] = { osg * gnx[1], osg * gnx[2], osg * gnx[3] };
    double rex[NCOMP][NDOF];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c)
#pragma unroll
      for (int k = 0; k < NDOF; ++k) rex[c][k] = rnx[c][k];
    {
      // prefetch the next task of this lane
      const int an = (q == 0) ? ta[1] : (q == 1) ? ta[2] : (q == 2) ? ta[3] : -1;
      const int fq = (q == 0) ? tf[1] : (q == 1) ? tf[2] : tf[3];
      const int nq = (q == 0) ? tn[1] : (q == 1) ? tn[2] : tn[3];
      if (an >= 0) {
        load_row<4>(geo, fq, gnx);
#ifndef QDG_KO_EXT
        if (TASK_KIND(an) == TASK_EXT) load_row<NPROP>(U, nq, &rnx[0][0]);
#endif
      }
    }
    const bool bnd = kind == TASK_BND;

    // vertex states of both tets at the three face vertices
    double Vo[3][NCOMP], Vn[3][NCOMP];
    int no[3], nn[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { no[j] = lpofa(lf, j); nn[j] = (code >> (2 * j)) & 3; }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
#ifdef QDG_KO_LDSRD
      for (int c = 0; c < NCOMP; ++c) Vo[j][c] = 1.0 + 0.01 * (c + j) + area;
#else
      for (int c = 0; c < NCOMP; ++c) Vo[j][c] = nod[LIDX(el, no[j], c)];
#endif
    }
    if (kind == TASK_INT) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
#pragma unroll
#ifdef QDG_KO_LDSRD
        for (int c = 0; c < NCOMP; ++c) Vn[j][c] = 1.1 + 0.01 * (c + j) + area;
#else
        for (int c = 0; c < NCOMP; ++c) Vn[j][c] = nod[LIDX(pl, nn[j], c)];
#endif
      }
    } else if (kind == TASK_EXT) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double b1, b2, b3;
        vertex_basis(nn[j], b1, b2, b3);
#pragma unroll
        for (int c = 0; c < NCOMP; ++c)
          Vn[j][c] = rex[c][0] + rex[c][1] * b1 + rex[c][2] * b2 + rex[c][3] * b3;
      }
    } else {
      // Extrapolate: u_r = u_l; Symmetry: mirrored momentum (DGCompFlow.hpp:672-690),
      // a linear map, applied to the vertex states
      const double refl = (bc == 2) ? 2.0 : 0.0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double vn2 = refl * (Vo[j][1] * fn[0] + Vo[j][2] * fn[1] + Vo[j][3] * fn[2]);
        Vn[j][0] = Vo[j][0];
        Vn[j][1] = Vo[j][1] - vn2 * fn[0];
        Vn[j][2] = Vo[j][2] - vn2 * fn[1];
        Vn[j][3] = Vo[j][3] - vn2 * fn[2];
        Vn[j][4] = Vo[j][4];
      }
    }
    const double wsel = (bnd && bc == 0) ? 0.0 : 1.0;   // boundary face without a BC: no flux

    // The 3-point rule (Quadrature.cpp:261-339) puts weight 2/3 on one face vertex and 1/6
    // on the other two, all three points weigh 1/3: with B = (V0+V1+V2)/6 the state at point g
    // is B + V_h(g)/2, h(g) = (g+1)%3, and the vertex-weighted flux sums are
    //   W_j = A/18 (F_0+F_1+F_2) + A/6 F_g(j),  g(j) = (j+2)%3
    // -- one FMA per state component and no arithmetic on the fluxes inside the point loop.
    double Bo[NCOMP], Bn[NCOMP], Fg[3][NCOMP], dsum = 0.0;
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      Bo[c] = (Vo[0][c] + Vo[1][c] + Vo[2][c]) * (1.0 / 6.0);
      Bn[c] = (Vn[0][c] + Vn[1][c] + Vn[2][c]) * (1.0 / 6.0);
    }
    [[maybe_unused]] ElemGeom gdir;
    if constexpr (HAS_DIRICHLET) {
      if (bnd && bc == 1) load_geom(m, tile_e0 + el, gdir);
    }
#pragma unroll
    for (int ig = 0; ig < NGF; ++ig) {
      constexpr int H[3] = { 1, 2, 0 };
      const int h = H[ig];
      double so[NCOMP], sn[NCOMP];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        so[c] = fma(0.5, Vo[h][c], Bo[c]);
        sn[c] = fma(0.5, Vn[h][c], Bn[c]);
      }
      if constexpr (HAS_DIRICHLET) {
        if (bnd && bc == 1) {
          double P[3];
          face_point(gdir, lf, T.fs[ig][0], T.fs[ig][1], T.fs[ig][2], P);
          prob_solution<PROB>(ph, P[0], P[1], P[2], t, sn);
        }
      }
#ifdef QDG_KO_GP
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) Fg[ig][c] = so[c] + sn[c] * fn[c % 3];
      continue;
#endif
      Prim qo, qn;
      primitives(ph, fn, so, qo);
      primitives(ph, fn, sn, qn);
      if (WITH_DT) {
        // delt += std::max(dSV_l, dSV_r) = (a < b) ? b : a in STORED (left, right) order;
        // boundary faces: dSV_r = 0
        const double d_o = fabs(qo.vn) + qo.a;
        const double d_n = bnd ? 0.0 : fabs(qn.vn) + qn.a;
        const bool take_n = own_left ? (d_o < d_n) : !(d_n < d_o);
        dsum += take_n ? d_n : d_o;
      }
      if (ph.flux == 1) {
        // Lax-Friedrichs is symmetric under the mirror image (LaxFriedrichs.hpp:34-88)
        flux_lf_q(fn, so, sn, qo, qn, Fg[ig]);
      } else {
        const double rlr = fast_sqrt(sn[0] * qo.ir);
        const double irlr1 = fast_rcp(1.0 + rlr);
        const double vnroe = (qn.vn * rlr + qo.vn) * irlr1;
        const double aroe = (qn.a * rlr + qo.a) * irlr1;
        const double Sl = fmin(qo.vn - qo.a, vnroe - aroe);
        const double Sr = fmax(qn.vn + qn.a, vnroe + aroe);
        const double ml = so[0] * (Sl - qo.vn), mr = sn[0] * (Sr - qn.vn);
        const double Sm = (mr * qn.vn - ml * qo.vn + qo.p - qn.p) * fast_rcp(mr - ml);
        const double pStar = so[0] * (qo.vn - Sl) * (qo.vn - Sm) + qo.p;
        // stored orientation = own frame:  Sl>0: own | Sl<=0,Sm>0: own* | Sm<=0,Sr>=0: nbr* | else nbr
        // mirrored:                        Sr<0: nbr | Sr>=0,Sm<0: nbr* | Sm>=0,Sl<=0: own* | else own
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
        Fg[ig][0] = sv * u0;
        Fg[ig][1] = sv * u1 + dp * fn[0];
        Fg[ig][2] = sv * u2 + dp * fn[1];
        Fg[ig][3] = sv * u3 + dp * fn[2];
        Fg[ig][4] = sv * u4 + e4;
      }
#if QDG_TILE_GP_SERIAL
      __builtin_amdgcn_sched_barrier(0);     // keep the three points in sequence (register pressure)
#endif
    }

    // ---- scatter the vertex-weighted flux sums: the own tet loses, the neighbour gains ---
    {
      const double k1 = area * wsel * (1.0 / 18.0), k2 = area * wsel * (1.0 / 6.0);
      double W[3][NCOMP];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const double Ssum = k1 * ((Fg[0][c] + Fg[1][c]) + Fg[2][c]);
        W[0][c] = fma(k2, Fg[2][c], Ssum);
        W[1][c] = fma(k2, Fg[0][c], Ssum);
        W[2][c] = fma(k2, Fg[1][c], Ssum);
      }
#ifdef QDG_KO_ATOM
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) ko_sum += W[j][c];
      continue;
#endif
#pragma unroll
      for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c)
          __hip_atomic_fetch_add(accN + LIDX(el, no[j], c), -W[j][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (WITH_DT) {
        dsum *= area * (1.0 / 3.0);
        __hip_atomic_fetch_add(sdelt + el, dsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (kind == TASK_INT) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
          for (int c = 0; c < NCOMP; ++c)
            __hip_atomic_fetch_add(accN + LIDX(pl, nn[j], c), W[j][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (WITH_DT) __hip_atomic_fetch_add(sdelt + pl, dsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
#ifdef QDG_KO_ATOM
  if (ko_sum == 1234.56789) accN[tid] = ko_sum;
#endif
  // phase-2 inputs are requested before the barrier (their latency overlaps the
  // other waves' last tasks)
  double u[NCOMP][NDOF], un[NCOMP][NDOF];
  double vol = 1.0;
  ElemGeom g;
  if (tid < nloc) {
    const int e = tile_e0 + tid;
    load_row<NPROP>(U, e, &u[0][0]);          // modal row again
    if (FUSE_RK) load_row<NPROP>(Un, e, &un[0][0]);
    vol = m.vol[e];
    const int stride = m.stride;
    const int n0 = m.inpoel[e], n1 = m.inpoel[(size_t)stride + e], n2 = m.inpoel[(size_t)2 * stride + e],
              n3 = m.inpoel[(size_t)3 * stride + e];
    double q[4];
    load_row<4>(m.xyz4, n0, q); g.p[0][0] = q[0]; g.p[0][1] = q[1]; g.p[0][2] = q[2];
    load_row<4>(m.xyz4, n1, q); g.p[1][0] = q[0]; g.p[1][1] = q[1]; g.p[1][2] = q[2];
    load_row<4>(m.xyz4, n2, q); g.p[2][0] = q[0]; g.p[2][1] = q[1]; g.p[2][2] = q[2];
    load_row<4>(m.xyz4, n3, q); g.p[3][0] = q[0]; g.p[3][1] = q[1]; g.p[3][2] = q[2];
  }
  __syncthreads();

  // ---- phase 2: one lane per tet: volume (+source) term, epilogue, store --------
  double dte = DBL_MAX;
  double acc[NCOMP][NDOF];
  if (tid < nloc) {
    {
      // R[c][k] = sum_v accN[v][c] * B_k(vertex v)
      double nv[4][NCOMP];
#pragma unroll
      for (int vx = 0; vx < 4; ++vx)
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) nv[vx][c] = accN[LIDX(tid, vx, c)];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        acc[c][0] = (nv[0][c] + nv[1][c]) + (nv[2][c] + nv[3][c]);
        acc[c][1] = nv[1][c] - nv[0][c];
        acc[c][2] = 2.0 * nv[2][c] - nv[0][c] - nv[1][c];
        acc[c][3] = 3.0 * nv[3][c] - nv[0][c] - nv[1][c] - nv[2][c];
      }
    }
#ifndef QDG_KO_P2VOL
    {
      double ji[3][3];
      inverse_jacobian(g, ji);
      double Fs[NCOMP][3];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) Fs[c][0] = Fs[c][1] = Fs[c][2] = 0.0;
#pragma unroll
      for (int ig = 0; ig < NGV; ++ig) {
        double s[NCOMP];
        state_from<NDOF>(u, T.vB[ig], s);
        const double ir = fast_rcp(s[0]);
        const double uu = s[1] * ir, vv = s[2] * ir, ww = s[3] * ir;
        const double p = eos_pressure(ph, s[0], uu, vv, ww, s[4]);
        const double wg = T.vw[ig];
        const double h = s[4] + p;
        Fs[0][0] += wg * s[1];            Fs[0][1] += wg * s[2];            Fs[0][2] += wg * s[3];
        Fs[1][0] += wg * (s[1] * uu + p); Fs[1][1] += wg * (s[2] * uu);     Fs[1][2] += wg * (s[3] * uu);
        Fs[2][0] += wg * (s[1] * vv);     Fs[2][1] += wg * (s[2] * vv + p); Fs[2][2] += wg * (s[3] * vv);
        Fs[3][0] += wg * (s[1] * ww);     Fs[3][1] += wg * (s[2] * ww);     Fs[3][2] += wg * (s[3] * ww + p);
        Fs[4][0] += wg * (uu * h);        Fs[4][1] += wg * (vv * h);        Fs[4][2] += wg * (ww * h);
      }
#pragma unroll
      for (int k = 1; k < NDOF; ++k) {
        const double g0 = T.vdB[0][0][k], g1 = T.vdB[0][1][k], g2 = T.vdB[0][2][k];
        const double dx = vol * (g0 * ji[0][0] + g1 * ji[1][0] + g2 * ji[2][0]);
        const double dy = vol * (g0 * ji[0][1] + g1 * ji[1][1] + g2 * ji[2][1]);
        const double dz = vol * (g0 * ji[0][2] + g1 * ji[1][2] + g2 * ji[2][2]);
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) acc[c][k] += Fs[c][0] * dx + Fs[c][1] * dy + Fs[c][2] * dz;
      }
    }
#endif
    if constexpr (prob_has_source<PROB>()) {
      const int ngs = NGV;
#pragma unroll 1
      for (int ig = 0; ig < ngs; ++ig) {
        constexpr bool one = false;
        const double xi = one ? 0.25 : T.vc[ig][0], eta = one ? 0.25 : T.vc[ig][1],
                     zeta = one ? 0.25 : T.vc[ig][2];
        const double w0 = 1.0 - xi - eta - zeta;
        double P[3], s[NCOMP];
#pragma unroll
        for (int d = 0; d < 3; ++d)
          P[d] = g.p[0][d] * w0 + g.p[1][d] * xi + g.p[2][d] * eta + g.p[3][d] * zeta;
        prob_src<PROB>(ph, P[0], P[1], P[2], t, s);
        const double wt = (one ? 1.0 : T.vw[ig]) * vol;
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
          const double ws = wt * s[c];
          acc[c][0] += ws;
          if (!one) {
#pragma unroll
            for (int k = 1; k < NDOF; ++k) acc[c][k] += ws * T.vB[ig][k];
          }
        }
      }
    }
    if (FUSE_RK) {
      constexpr double imf[4] = { 1.0, 10.0, 10.0 / 3.0, 5.0 / 3.0 };
      const double dtv = dtp[0] / vol;
#pragma unroll
      for (int c = 0; c < NCOMP; ++c)
#pragma unroll
        for (int k = 0; k < NDOF; ++k)
          acc[c][k] = rk_a * un[c][k] + rk_b * (u[c][k] + dtv * imf[k] * acc[c][k]);
    }
    if (WITH_DT) dte = vol / sdelt[tid];
  }
  // rows out, coalesced: a lane storing its own 160-B row issues 64 separate 16-B write
  // requests per wave instruction (3.4 TB/s measured, tools/ubench_rowstream.hip); the tile's
  // rows are one contiguous span, so they go through LDS (row-major over the vertex states,
  // which nobody reads any more) and leave as 1-KiB wave stores (6.1 TB/s)
  __syncthreads();
  if (tid < nloc) {
    double2* row = reinterpret_cast<double2*>(nod + (size_t)tid * NPROP);
#pragma unroll
    for (int j = 0; j < NPROP / 2; ++j) row[j] = make_double2((&acc[0][0])[2 * j], (&acc[0][0])[2 * j + 1]);
  }
  __syncthreads();
  {
    const double2* src = reinterpret_cast<const double2*>(nod);
    double2* dst = reinterpret_cast<double2*>(R + (size_t)tile_e0 * NPROP);
    const int nvalid = nloc * (NPROP / 2);
#pragma unroll
    for (int j = 0; j < NPROP / 2; ++j) {
      const int i = j * TILE_BS + tid;
      if (i < nvalid) dst[i] = src[i];
    }
  }

  if (WITH_DT) {
    for (int off = 32; off > 0; off >>= 1) dte = fmin(dte, __shfl_down(dte, off, 64));
    __shared__ double wmin[TILE_BS / 64];
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) wmin[wv] = dte;
    __syncthreads();
    if (tid == 0) {
      double mn = wmin[0];
      for (int w = 1; w < TILE_BS / 64; ++w) mn = fmin(mn, wmin[w]);
      blockmin[tile] = mn;
    }
  }
}

// ------------------------------------------------------------- limiters
// Superbee_P1, src/PDE/Limiter.cpp:155-316: only neighbour MEANS are read, so
// the in-place update is order independent.
// Limiter.cpp:283-301: phi = min over the face points of f(phi_gp),
// phi_gp = min(1, (uMax|uMin - u0) / (2 uNeg)), f(p) = max(0, max(min(2p,1), min(p,2))).
// f is non-decreasing, so phi = f(min phi_gp): the smallest ratio a/b is tracked by
// cross-multiplication (a, b >= 0) and divided once per component instead of once
// per point (|uNeg| <= 1e-14 -> phi_gp = 1: skipped).
template <int NDOF>
__device__ __forceinline__ void superbee_phi(const Tables<NDOF>& T, const double (&u)[NCOMP][NDOF],
                                             const double* uMin, const double* uMax, double* phi)
{
  constexpr int NGF = Tables<NDOF>::NGF;
  double ra[NCOMP], rb[NCOMP];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) { ra[c] = 1.0; rb[c] = 1.0; }
#pragma unroll 1
  for (int lf = 0; lf < 4; ++lf)
#pragma unroll
    for (int ig = 0; ig < NGF; ++ig) {
      double s[NCOMP];
      state_from<NDOF>(u, T.fB[lf][ig], s);
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const double uNeg = s[c] - u[c][0];
        const double a = (uNeg > 0.0) ? (uMax[c] - u[c][0]) : (u[c][0] - uMin[c]);
        const double b = 2.0 * fabs(uNeg);
        const bool take = (fabs(uNeg) > 1.0e-14) && (a * rb[c] < ra[c] * b);
        ra[c] = take ? a : ra[c];
        rb[c] = take ? b : rb[c];
      }
    }
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) {
    const double pg = fmin(1.0, ra[c] * fast_rcp(rb[c]));
    phi[c] = fmax(0.0, fmax(fmin(2.0 * pg, 1.0), fmin(pg, 2.0)));
  }
}

template <int NDOF>
__global__ __launch_bounds__(256) void k_superbee(DevMesh m, double* __restrict__ U)
{
  if constexpr (NDOF > 1) {
    const Tables<NDOF>& T = tab<NDOF>();
    constexpr int NGF = Tables<NDOF>::NGF;
    constexpr int NPROP = NCOMP * NDOF;
    __shared__ double lds[256 * NPROP];
    const int tile_e0 = (m.blk0 + xcd_tile(blockIdx.x, gridDim.x)) * 256;
    const int e0 = tile_e0 + threadIdx.x;
    const bool active = e0 < m.nie;
    const int e = active ? e0 : m.nie - 1;
    const int stride = m.stride;
    double u[NCOMP][NDOF];
    // the lane reads its own row directly (5.7 TB/s measured for that access, against 4.5 for the
    // detour through LDS: tools/ubench_rowstream.hip); only the MEANS go to LDS, for the
    // neighbours in the tile (93 -> 85 us at 1 M tets)
    load_row<NPROP>(U, e, &u[0][0]);
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) lds[(size_t)threadIdx.x * NPROP + c * NDOF] = u[c][0];
    __syncthreads();
    double uMin[NCOMP], uMax[NCOMP], phi[NCOMP];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) { uMin[c] = uMax[c] = u[c][0]; phi[c] = 1.0; }
#pragma unroll
    for (int lf = 0; lf < 4; ++lf) {
      const int nb = m.nbr[(size_t)lf * stride + e];
      if (nb < 0) continue;
      const int r = nb - tile_e0;
      // in-tile neighbour: means from LDS (a ghost id may fall into the id range of a
      // ragged last tile: ghosts always come from global memory)
      if (nb < m.nie && (unsigned)r < 256u) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
          const double v = lds[(size_t)r * NPROP + c * NDOF];
          uMin[c] = fmin(uMin[c], v);
          uMax[c] = fmax(uMax[c], v);
        }
      } else {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
          const double v = U[fidx(c * NDOF, nb, NPROP)];
          uMin[c] = fmin(uMin[c], v);
          uMax[c] = fmax(uMax[c], v);
        }
      }
    }
    superbee_phi<NDOF>(T, u, uMin, uMax, phi);
    if (m.ndofel && m.ndofel[e] == 1) {    // pdg: P0 elements are not limited (Limiter.cpp:179-180)
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) phi[c] = 1.0;
    }
#pragma unroll
    for (int c = 0; c < NCOMP; ++c)
#pragma unroll
      for (int k = 1; k < 4; ++k) u[c][k] = phi[c] * u[c][k];
    // Out-of-tile neighbours may be read from U while another tile has already
    // stored its limited rows: safe, Superbee never changes a mean.
    tile_store_rows<NPROP>(U, tile_e0, m.nie, lds, &u[0][0]);
  }
}

// Stage-0 RK update fused with the limiter of stage 1 (DG-P1 CompFlow, Superbee):
//   Uout = Superbee( U0 + dt * R / L )        (DG.cpp:1478-1488 with a = 0, b = 1, then Limiter.cpp:155-316)
// U1 never goes to memory unlimited: one kernel reads U0 and R and writes the limited
// U1 (saves a full write + read of the state per time step).  Means of neighbours outside
// the 256-row tile are formed on the fly from U0 and R; ghost neighbours (rows >= nie) are
// read from Uout, where the halo exchange has already put the owners' U1.
template <int NDOF>
__global__ __launch_bounds__(256) void k_upd_superbee(DevMesh m, const double* __restrict__ dtp,
                                                      const double* __restrict__ U0,
                                                      const double* __restrict__ R,
                                                      double* __restrict__ Uout)
{
  static_assert(NDOF == 4, "fused update + Superbee exists for DG-P1");
  const Tables<NDOF>& T = tab<NDOF>();
  constexpr int NPROP = NCOMP * NDOF, NCH = NPROP / 2;     // 16-byte chunks per row
  __shared__ double lds[256 * NPROP];
  __shared__ double sdtv[256];
  const int tid = threadIdx.x;
  const int tile_e0 = (m.blk0 + xcd_tile(blockIdx.x, gridDim.x)) * 256;
  const int e0 = tile_e0 + tid;
  const bool active = e0 < m.nie;
  const int e = active ? e0 : m.nie - 1;
  const int stride = m.stride;
  const double dt = dtp[0];
  sdtv[tid] = dt / m.vol[e];                       // the row's dt / vol, as k_rk forms it
  __syncthreads();
  // the tile's rows of U1 = U0 + dt R / L go to LDS in one coalesced pass over both arrays
  {
    const double2* su = reinterpret_cast<const double2*>(U0 + (size_t)tile_e0 * NPROP);
    const double2* sr = reinterpret_cast<const double2*>(R + (size_t)tile_e0 * NPROP);
    double2* dst = reinterpret_cast<double2*>(lds);
    const int nvalid = (m.nie - tile_e0 < 256 ? m.nie - tile_e0 : 256) * NCH;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int i = j * 256 + tid;
      double2 v = make_double2(1.0, 1.0);
      if (i < nvalid) {
        const double2 a = su[i], b = sr[i];
        const int row = i / NCH, hi = (i - row * NCH) & 1;     // chunk holds modes (0,1) or (2,3)
        const double dtv = sdtv[row];
        const double f0 = hi ? 10.0 / 3.0 : 1.0, f1 = hi ? 5.0 / 3.0 : 10.0;
        v = make_double2(a.x + dtv * f0 * b.x, a.y + dtv * f1 * b.y);
      }
      dst[i] = v;
    }
  }
  __syncthreads();
  double u[NCOMP][NDOF];
  lds_row<NPROP>(lds, tid, &u[0][0]);
  double uMin[NCOMP], uMax[NCOMP], phi[NCOMP];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) { uMin[c] = uMax[c] = u[c][0]; phi[c] = 1.0; }
#pragma unroll
  for (int lf = 0; lf < 4; ++lf) {
    const int nb = m.nbr[(size_t)lf * stride + e];
    if (nb < 0) continue;
    const int rr = nb - tile_e0;
    if (nb < m.nie && (unsigned)rr < 256u) {     // (a ghost id may alias the ragged last tile)
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const double v = lds[(size_t)rr * NPROP + c * NDOF];
        uMin[c] = fmin(uMin[c], v); uMax[c] = fmax(uMax[c], v);
      }
    } else if (nb >= m.nie) {                   // ghost: the owner's U1 mean, already exchanged
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const double v = Uout[fidx(c * NDOF, nb, NPROP)];
        uMin[c] = fmin(uMin[c], v); uMax[c] = fmax(uMax[c], v);
      }
    } else {
      const double dtn = dt / m.vol[nb];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const double v = U0[fidx(c * NDOF, nb, NPROP)] + dtn * 1.0 * R[fidx(c * NDOF, nb, NPROP)];
        uMin[c] = fmin(uMin[c], v); uMax[c] = fmax(uMax[c], v);
      }
    }
  }
  superbee_phi<NDOF>(T, u, uMin, uMax, phi);
#pragma unroll
  for (int c = 0; c < NCOMP; ++c)
#pragma unroll
    for (int k = 1; k < 4; ++k) u[c][k] = phi[c] * u[c][k];
  tile_store_rows<NPROP>(Uout, tile_e0, m.nie, lds, &u[0][0]);
}

// send side of the same fusion: slab row j = U0[e] + dt * R[e] / L[e], e = send_elem[j]
__global__ __launch_bounds__(256) void k_halo_pack_upd(const double* __restrict__ U0,
                                                       const double* __restrict__ R,
                                                       const double* __restrict__ dtp,
                                                       const double* __restrict__ vol,
                                                       const int* __restrict__ send_elem, int nsend,
                                                       double* __restrict__ slab)
{
  constexpr int NPROP = NCOMP * 4;
  constexpr double imf[4] = { 1.0, 10.0, 10.0 / 3.0, 5.0 / 3.0 };
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nsend * NPROP) return;
  const int j = i / NPROP, p = i - j * NPROP, k = p & 3;
  const int e = send_elem[j];
  const double dtv = dtp[0] / vol[e];
  const double f = (k == 0) ? imf[0] : (k == 1) ? imf[1] : (k == 2) ? imf[2] : imf[3];
  slab[i] = U0[(size_t)e * NPROP + p] + dtv * f * R[(size_t)e * NPROP + p];
}

// WENO_P1, src/PDE/Limiter.cpp:29-153 (Jacobi: reads Uin, writes modes 1-3 of
// Uout; all other planes are copied by the caller)
template <int NDOF, int BS>
__global__ __launch_bounds__(BS) void k_weno(DevMesh m, double cweight,
                                             const double* __restrict__ Uin,
                                             double* __restrict__ Uout)
{
  // every row of Uout is written here (own row with modes 1-3 replaced; ghost rows
  // copied), so the Jacobi sweep needs no separate copy of the state.  The kernel is bound by
  // the number of scattered lane addresses its memory instructions carry, so a neighbour's
  // three gradient modes of a component (24 contiguous bytes) come as one 8-byte and one
  // 16-byte load, and the rows leave through LDS as coalesced wave stores.
  constexpr int NPROP = NCOMP * NDOF;
  __shared__ __attribute__((aligned(16))) double stage[NDOF > 1 ? BS * NPROP : 2];
  const int blk = xcd_tile(blockIdx.x, gridDim.x);
  const int e0 = blk * BS + threadIdx.x;
  const bool active = e0 < m.ne;
  const int e = active ? e0 : m.ne - 1;
  double r[NCOMP][NDOF];
  load_row<NPROP>(Uin, e, &r[0][0]);
  if constexpr (NDOF > 1) {
    if (e < m.nie) {
      const int stride = m.stride;
      int nb[4];
#pragma unroll
      for (int lf = 0; lf < 4; ++lf) nb[lf] = m.nbr[(size_t)lf * stride + e];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        double g[5][3], wd[5], wtot = 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d) g[0][d] = r[c][1 + d];
#pragma unroll
        for (int is = 1; is < 5; ++is) {
          const int n = nb[is - 1];
          g[is][0] = g[is][1] = g[is][2] = 0.0;
          if (n >= 0) {
            // modes 1, 2, 3 of component c: doubles c*NDOF + 1 .. + 3 of the row; the row is
            // 16-byte aligned and c*NDOF is even, so the pair (2, 3) is a 16-byte load
            const double* pn = Uin + (size_t)n * NPROP + c * NDOF + 1;
            const double2 v = *reinterpret_cast<const double2*>(__builtin_assume_aligned(pn + 1, 16));
            g[is][0] = pn[0]; g[is][1] = v.x; g[is][2] = v.y;
          }
        }
#pragma unroll
        for (int is = 0; is < 5; ++is) {
          const double wst = (is == 0) ? cweight : (nb[is - (is > 0)] >= 0 ? 1.0 : 0.0);
          const double osc = sqrt(g[is][0] * g[is][0] + g[is][1] * g[is][1] + g[is][2] * g[is][2]);
          const double q = 1.0e-8 + osc;
          wd[is] = wst / (q * q);
          wtot += wd[is];
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          double a = 0.0;
#pragma unroll
          for (int is = 0; is < 5; ++is) a += (wd[is] / wtot) * g[is][d];
          r[c][1 + d] = a;
        }
      }
    }
    {
      double2* row = reinterpret_cast<double2*>(stage + (size_t)threadIdx.x * NPROP);
#pragma unroll
      for (int j = 0; j < NPROP / 2; ++j) row[j] = make_double2((&r[0][0])[2 * j], (&r[0][0])[2 * j + 1]);
      __syncthreads();
      const int r0 = blk * BS;
      const int nrow = (m.ne - r0 < BS) ? m.ne - r0 : BS;
      const double2* src = reinterpret_cast<const double2*>(stage);
      double2* dst = reinterpret_cast<double2*>(Uout + (size_t)r0 * NPROP);
      const int nvalid = nrow * (NPROP / 2);
#pragma unroll 5
      for (int j = 0; j < NPROP / 2; ++j) {
        const int i = j * BS + threadIdx.x;
        if (i < nvalid) dst[i] = src[i];
      }
    }
  } else {
    if (active) store_row<NPROP>(Uout, e, &r[0][0]);
  }
}

// ------------------------------------------------------------- time step
// dg::CompFlow::dt, src/PDE/CompFlow/DGCompFlow.hpp:206-406, element-centric:
// delt[e] = sum over own faces and Gauss points of max(dSV_own, dSV_nbr),
// dSV = wt*(|vn|+a); returns per-block minima of vol/delt.
template <int NDOF>
__global__ __launch_bounds__(256) void k_dt(DevMesh m, Phys ph, const double* __restrict__ U,
                                            double* __restrict__ blockmin)
{
  const int e = xcd_tile(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  double dte = DBL_MAX;
  if (e < m.nie) {
    const Tables<NDOF>& T = tab<NDOF>();
    constexpr int NGF = Tables<NDOF>::NGF;
    const int stride = m.stride;
    double delt = 0.0;
#pragma unroll 1
    for (int lf = 0; lf < 4; ++lf) {
      const int nb = m.nbr[(size_t)lf * stride + e];
      const int info = m.finfo[(size_t)lf * stride + e];
      const int f = m.fid[(size_t)lf * stride + e];
      const double area = m.farea[f];
      const double fn[3] = { m.fnx[f], m.fny[f], m.fnz[f] };
      const bool own_left = (info >> 6) & 1;
#pragma unroll 1
      for (int ig = 0; ig < NGF; ++ig) {
        const double wt = T.fw[ig] * area;
        double s[NCOMP];
        state_gather<NDOF>(U, stride, e, T.fB[lf][ig], s);
        double rho = s[0], u = s[1] / rho, v = s[2] / rho, w = s[3] / rho;
        double p = eos_pressure(ph, rho, u, v, w, s[4]);
        double a = eos_soundspeed(ph, rho, p);
        double vn = u * fn[0] + v * fn[1] + w * fn[2];
        const double dl = wt * (fabs(vn) + a);
        double dr = 0.0;
        if (nb >= 0) {
          double xi, eta, zeta, Bn[NDOF];
          nbr_ref_coords(info, T.fs[ig][0], T.fs[ig][1], T.fs[ig][2], xi, eta, zeta);
          eval_basis<NDOF>(xi, eta, zeta, Bn);
          state_gather<NDOF>(U, stride, nb, Bn, s);
          rho = s[0]; u = s[1] / rho; v = s[2] / rho; w = s[3] / rho;
          p = eos_pressure(ph, rho, u, v, w, s[4]);
          a = eos_soundspeed(ph, rho, p);
          vn = u * fn[0] + v * fn[1] + w * fn[2];
          dr = wt * (fabs(vn) + a);
        }
        // std::max(dSV_l, dSV_r) with (face-left, face-right) argument order,
        // i.e. (a < b) ? b : a -- keeps the reference's NaN behaviour
        const double a_ = own_left ? dl : dr, b_ = own_left ? dr : dl;
        delt += (a_ < b_) ? b_ : a_;
      }
    }
    dte = m.vol[e] / delt;
  }
  // wave reduction (64 lanes), then across the 4 waves through LDS
  for (int off = 32; off > 0; off >>= 1) dte = fmin(dte, __shfl_down(dte, off, 64));
  __shared__ double wmin[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wmin[wv] = dte;
  __syncthreads();
  if (threadIdx.x == 0)
    blockmin[blockIdx.x] = fmin(fmin(wmin[0], wmin[1]), fmin(wmin[2], wmin[3]));
}

// final reduction of the block minima by one workgroup; applies the CFL scaling
// dt = min * cfl/(2p+1) (src/Inciter/DG.cpp:1404-1418) and the cap to `tleft`
__global__ __launch_bounds__(256) void k_dt_final(const double* __restrict__ blockmin, int n,
                                                  double scale, double tleft,
                                                  double* __restrict__ out_raw,
                                                  double* __restrict__ out_dt)
{
  double v = DBL_MAX;
  for (int i = threadIdx.x; i < n; i += blockDim.x) v = fmin(v, blockmin[i]);
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
  __shared__ double wmin[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wmin[wv] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double mn = fmin(fmin(wmin[0], wmin[1]), fmin(wmin[2], wmin[3]));
    out_raw[0] = mn;
    out_dt[0] = fmin(mn * scale, tleft);
  }
}

// ------------------------------------------------------------- RK update
// src/Inciter/DG.cpp:39-40,1478-1488 with L = vol*massfac[k] (Mass.cpp:25-73)
// recomputed instead of streamed.  One lane per (plane, element).
template <int NDOF>
__global__ __launch_bounds__(256) void k_rk(DevMesh m, double a, double b,
                                            const double* __restrict__ dt,
                                            const double* __restrict__ Un,
                                            const double* __restrict__ R, const double* U,
                                            double* Uout)
{
  // flat, fully coalesced sweep over the nie*NPROP doubles of the interior rows
  constexpr int NPROP = NCOMP * NDOF;
  constexpr double imf[10] = { 1.0, 10.0, 10.0 / 3.0, 5.0 / 3.0, 35.0, 21.0, 14.0, 7.0,
                               14.0 / 3.0, 7.0 / 3.0 };
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)m.nie * NPROP) return;
  const int e = (int)(i / NPROP);
  const int k = (int)(i - (size_t)e * NPROP) % NDOF;
  // imf[k] through selects (no runtime-indexed array)
  double f = imf[0];
#pragma unroll
  for (int j = 1; j < NDOF; ++j) f = (k == j) ? imf[j] : f;
  const double dtv = dt[0] / m.vol[e];
  Uout[i] = a * Un[i] + b * (U[i] + dtv * f * R[i]);   // Uout may alias U (in place)
}

// ------------------------------------------------------------- setup ops
// tk::mass, src/PDE/Integrate/Mass.cpp:25-73 (written straight into the
// caller's AoS layout on the host side; here SoA planes for all ne rows)
template <int NDOF>
__global__ void k_mass(DevMesh m, double* __restrict__ L)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m.ne) return;
  const double vol = m.vol[e];
  const double f[10] = { vol, vol / 10.0, vol * 3.0 / 10.0, vol * 3.0 / 5.0, vol / 35.0,
                         vol / 21.0, vol / 14.0, vol / 7.0, vol * 3.0 / 14.0, vol * 3.0 / 7.0 };
#pragma unroll
  for (int c = 0; c < NCOMP; ++c)
#pragma unroll
    for (int k = 0; k < NDOF; ++k) L[fidx(c * NDOF + k, e, NCOMP * NDOF)] = f[k];
}

// tk::initialize, src/PDE/Integrate/Initialize.cpp:29-201 (interior tets)
template <int NDOF, int PROB>
__global__ __launch_bounds__(256) void k_init(DevMesh m, Phys ph, double t, double* __restrict__ U)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m.nie) return;
  const QuadTet& Q = c_qinit[order_index<NDOF>()];
  ElemGeom g;
  load_geom(m, e, g);
  const double vol = m.vol[e];
  double acc[NCOMP][NDOF];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c)
#pragma unroll
    for (int k = 0; k < NDOF; ++k) acc[c][k] = 0.0;
#pragma unroll 1
  for (int ig = 0; ig < Q.ng; ++ig) {
    const double xi = Q.c[ig][0], eta = Q.c[ig][1], zeta = Q.c[ig][2];
    const double w0 = 1.0 - xi - eta - zeta;
    double P[3], s[NCOMP], B[NDOF];
#pragma unroll
    for (int d = 0; d < 3; ++d)
      P[d] = g.p[0][d] * w0 + g.p[1][d] * xi + g.p[2][d] * eta + g.p[3][d] * zeta;
    eval_basis<NDOF>(xi, eta, zeta, B);
    prob_solution<PROB>(ph, P[0], P[1], P[2], t, s);
    const double wt = Q.w[ig] * vol;
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      acc[c][0] += wt * s[c];
#pragma unroll
      for (int k = 1; k < NDOF; ++k) acc[c][k] += wt * s[c] * B[k];
    }
  }
  const double f[10] = { vol, vol / 10.0, vol * 3.0 / 10.0, vol * 3.0 / 5.0, vol / 35.0,
                         vol / 21.0, vol / 14.0, vol / 7.0, vol * 3.0 / 14.0, vol * 3.0 / 7.0 };
#pragma unroll
  for (int c = 0; c < NCOMP; ++c)
#pragma unroll
    for (int k = 0; k < NDOF; ++k) U[fidx(c * NDOF + k, e, NCOMP * NDOF)] = acc[c][k] / f[k];
}

// block reduction of the 15 diagnostics partials (sums 0..9, maxima 10..14) in a
// fixed order; one row of `part` per workgroup
__device__ __forceinline__ void diag_block_reduce(const double (&v)[15], double* __restrict__ part)
{
  __shared__ double sh[4][15];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 15; ++i) {
    double x = v[i];
    for (int off = 32; off > 0; off >>= 1) {
      const double y = __shfl_down(x, off, 64);
      x = (i < 10) ? x + y : fmax(x, y);
    }
    if (lane == 0) sh[wv][i] = x;
  }
  __syncthreads();
  if (threadIdx.x < 15) {
    const int i = threadIdx.x;
    const double r = (i < 10) ? ((sh[0][i] + sh[1][i]) + (sh[2][i] + sh[3][i]))
                              : fmax(fmax(sh[0][i], sh[1][i]), fmax(sh[2][i], sh[3][i]));
    part[(size_t)blockIdx.x * 15 + i] = r;
  }
}

// ElemDiagnostics::compute_diag, src/Inciter/ElemDiagnostics.cpp:116-215.
// Per-block partial sums (deterministic two-pass reduction): 15 doubles/block.
template <int NDOF, int PROB>
__global__ __launch_bounds__(256) void k_diag(DevMesh m, Phys ph, double t_new,
                                              const double* __restrict__ U,
                                              double* __restrict__ part)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  double v[15];
#pragma unroll
  for (int i = 0; i < 15; ++i) v[i] = 0.0;
  if (e < m.nie) {
    // pdg: NGdiag(ndofel[e]) points and the element's own number of modes
    // (ElemDiagnostics.cpp:144,171,186)
    const bool p0 = NDOF > 1 && m.ndofel && m.ndofel[e] == 1;
    const QuadTet& Q = p0 ? c_qdiag[0] : c_qdiag[order_index<NDOF>()];
    ElemGeom g;
    load_geom(m, e, g);
    const double vol = m.vol[e];
#pragma unroll 1
    for (int ig = 0; ig < Q.ng; ++ig) {
      const double xi = Q.c[ig][0], eta = Q.c[ig][1], zeta = Q.c[ig][2];
      const double w0 = 1.0 - xi - eta - zeta;
      double P[3], s[NCOMP], u[NCOMP], B[NDOF];
#pragma unroll
      for (int d = 0; d < 3; ++d)
        P[d] = g.p[0][d] * w0 + g.p[1][d] * xi + g.p[2][d] * eta + g.p[3][d] * zeta;
      eval_basis<NDOF>(xi, eta, zeta, B);
      if (p0) {
#pragma unroll
        for (int k = 1; k < NDOF; ++k) B[k] = 0.0;
      }
      prob_solution<PROB>(ph, P[0], P[1], P[2], t_new, s);
      state_gather<NDOF>(U, m.stride, e, B, u);
      const double wt = Q.w[ig] * vol;
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const double d = u[c] - s[c];
        v[c] += wt * u[c] * u[c];
        v[5 + c] += wt * d * d;
        v[10 + c] = fmax(v[10 + c], fabs(d));
      }
    }
  }
  diag_block_reduce(v, part);
}

__global__ void k_diag_final(const double* __restrict__ part, int nblk, double* __restrict__ out)
{
  const int i = threadIdx.x;
  if (i >= 15) return;
  double r = 0.0;
  for (int b = 0; b < nblk; ++b) {
    const double y = part[(size_t)b * 15 + i];
    r = (i < 10) ? r + y : fmax(r, y);
  }
  out[i] = r;
}

// ================================================================ Problem::solution
// analytic / initial solution at arbitrary points (DGPDE::analyticSolution,
// src/PDE/DGPDE.hpp:141-144): out[i*ncomp + c]
template <int PROB>
__global__ __launch_bounds__(256) void k_solution(Phys ph, int n, const double* __restrict__ x,
                                                  const double* __restrict__ y,
                                                  const double* __restrict__ z, double t,
                                                  double* __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s[NCOMP];
  prob_solution<PROB>(ph, x[i], y[i], z[i], t, s);
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) out[(size_t)i * NCOMP + c] = s[c];
}

// ================================================================ field output
// Problem::fieldOutput as dg::CompFlow::fieldOutput calls it (DGCompFlow.hpp:447-462:
// V = 0, vol = geoElem(:,0), coord = element centroids geoElem(:,1..3)) -- every field
// of the Problem's own list, from the cell means:
//   SodShocktube.cpp:160-258 (Sedov, RotatedSod alike; 6 numerical fields),
//   VorticalFlow.cpp:156-254 (12: numerical/analytical interleaved; pressure_numerical is
//     evaluated with the ANALYTIC velocities -- the reference overwrites u,v,w first),
//   TaylorGreen.cpp:135-240 (15, three err(.) fields), NLEnergyGrowth.cpp:233-318 (14),
//   RayleighTaylor.cpp:223-314 (18; reads row entries 0..4, not c*rdof), UserDefined.cpp:105-169 (7).
// With V = 0 an err(.) field is x/0 (+inf, NaN where x == 0), as in the reference's goldens.
// Two callers: the resident state (device rows, centroid = mean of the 4 nodes with the
// association of tk::genGeoElemTet, output in the caller's numbering through d2h) and the
// stateless DGPDE::fieldOutput (caller's rows + the caller's geoElem).
template <int PROB> constexpr int prob_nfield()
{
  return PROB == 3 ? 12 : PROB == 4 ? 15 : PROB == 7 ? 14 : PROB == 10 ? 18 : PROB == 0 ? 7 : 6;
}

template <int PROB>
__global__ __launch_bounds__(256) void k_field_output(DevMesh m, Phys ph, int ndof, double t,
                                                      const double* __restrict__ U,
                                                      const double* __restrict__ geoElem, int nrows,
                                                      double* __restrict__ out)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nrows) return;
  size_t h;
  double vol, x, y, z;
  if (geoElem) {
    h = (size_t)e;
    vol = geoElem[4 * h]; x = geoElem[4 * h + 1]; y = geoElem[4 * h + 2]; z = geoElem[4 * h + 3];
  } else {
    h = (size_t)m.d2h[e];
    vol = m.vol[e];
    double p[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = m.inpoel[(size_t)i * m.stride + e];
      p[i][0] = m.x[n]; p[i][1] = m.y[n]; p[i][2] = m.z[n];
    }
    x = (p[0][0] + p[1][0] + p[2][0] + p[3][0]) / 4.0;
    y = (p[0][1] + p[1][1] + p[2][1] + p[3][1]) / 4.0;
    z = (p[0][2] + p[1][2] + p[2][2] + p[3][2]) / 4.0;
  }
  const size_t n = (size_t)nrows;
  const double* ue = U + (size_t)e * NCOMP * ndof;
  const double r = ue[0], ru = ue[ndof], rv = ue[2 * ndof], rw = ue[3 * ndof], re = ue[4 * ndof];
  const double V = 0.0;
#define OUT(f) out[(size_t)(f) * n + h]
  if constexpr (PROB == 3) {
    const double a = ph.alpha, b = ph.beta, p0 = ph.p0, g = ph.gamma;
    const double u = a * x - b * y, v = b * x + a * y, w = -2.0 * a * z;
    OUT(0) = r; OUT(1) = 1.0;
    OUT(2) = ru / r; OUT(3) = u;
    OUT(4) = rv / r; OUT(5) = v;
    OUT(6) = rw / r; OUT(7) = w;
    OUT(8) = re / r;
    OUT(9) = 0.5 * (u * u + v * v + w * w) + (p0 - 2.0 * a * a * z * z) / (g - 1.0);
    OUT(10) = eos_pressure(ph, r, u, v, w, re);
    OUT(11) = p0 - 2.0 * a * a * z * z;
  } else if constexpr (PROB == 4) {
    const double pi = 3.14159265358979323846;
    const double u = ru / r, v = rv / r, w = rw / r, E = re / r;
    const double ua = sin(pi * x) * cos(pi * y), va = -cos(pi * x) * sin(pi * y), wa = 0.0;
    const double Pa = 10.0 + r / 4.0 * (cos(2.0 * pi * x) + cos(2.0 * pi * y));
    const double Ea = eos_totalenergy(ph, r, ua / r, va / r, wa / r, Pa / r);
    OUT(0) = r; OUT(1) = 1.0;
    OUT(2) = u; OUT(3) = ua; OUT(4) = (ua - u) * (ua - u) * vol / V;
    OUT(5) = v; OUT(6) = va; OUT(7) = (va - v) * (va - v) * vol / V;
    OUT(8) = w; OUT(9) = wa;
    OUT(10) = E; OUT(11) = Ea; OUT(12) = (Ea - E) * (Ea - E) * vol / V;
    OUT(13) = eos_pressure(ph, r, u, v, w, r * E);
    OUT(14) = Pa;
  } else if constexpr (PROB == 7 || PROB == 10) {
    constexpr bool rt = PROB == 10;
    const double r_ = rt ? ue[0] : r;
    const double u = (rt ? ue[1] : ru) / r_, v = (rt ? ue[2] : rv) / r_, w = (rt ? ue[3] : rw) / r_,
                 E = (rt ? ue[4] : re) / r_;
    double s[NCOMP];
    prob_solution<PROB>(ph, x, y, z, t, s);
    const double ar = s[0], au = s[1] / s[0], av = s[2] / s[0], aw = s[3] / s[0], aE = s[4] / s[0];
    const double ap = eos_pressure(ph, ar, au, av, aw, ar * aE);
    OUT(0) = r_; OUT(1) = u; OUT(2) = v; OUT(3) = w; OUT(4) = E;
    OUT(5) = eos_pressure(ph, r_, u, v, w, r_ * E);
    OUT(6) = ar; OUT(7) = au; OUT(8) = av; OUT(9) = aw; OUT(10) = aE; OUT(11) = ap;
    OUT(12) = (r_ - s[0]) * (r_ - s[0]) * vol / V;
    OUT(13) = (E - aE) * (E - aE) * vol / V;
    if constexpr (rt) {
      const double ap0 = eos_pressure(ph, s[0], au, av, aw, s[4]);
      OUT(14) = (ap0 - ap) * (ap0 - ap) * vol / V;
      OUT(15) = (u - au) * (u - au) * vol / V;
      OUT(16) = (v - av) * (v - av) * vol / V;
      OUT(17) = (w - aw) * (w - aw) * vol / V;
    }
  } else {
    const double u = ru / r, v = rv / r, w = rw / r, E = re / r;
    OUT(0) = r; OUT(1) = u; OUT(2) = v; OUT(3) = w; OUT(4) = E;
    OUT(5) = eos_pressure(ph, r, u, v, w, r * E);
    if constexpr (PROB == 0) OUT(6) = ph.cv * (E - (u * u + v * v + w * w) / 2.0);
  }
#undef OUT
}

// per-element ndof of p-adaptive DG appended as one more element field
// (DG::writeFields, src/Inciter/DG.cpp:1201-1204)
__global__ __launch_bounds__(256) void k_field_ndof(DevMesh m, int nrows, double* __restrict__ out)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nrows) out[m.d2h[e]] = (double)m.ndofel[e];
}

// dg::CompFlow::avgElemToNode, src/PDE/CompFlow/DGCompFlow.hpp:465-552: every element's
// state at its four nodes (the P1 part of the basis, also for rdof = 10, :517-526; the
// reference coordinates of a node are 0/1, its Jacobian ratios :497-505), primitive
// quantities summed per node.  Caller's numbering (stateless call).  The sums are
// double atomics: the order of the ~20 contributions per node is not fixed (last-bit
// differences from run to run).
__global__ __launch_bounds__(256) void k_avg_elem_to_node(Phys ph, int rdof, int nelem, int nnode,
                                                          const int* __restrict__ inpoel,
                                                          const double* __restrict__ U,
                                                          double* __restrict__ out,
                                                          double* __restrict__ count)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nelem) return;
  const double* ue = U + (size_t)e * NCOMP * rdof;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double b1, b2, b3, s[NCOMP];
    vertex_basis(i, b1, b2, b3);
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      const double* uc = ue + c * rdof;
      s[c] = (rdof == 1) ? uc[0] : uc[0] + uc[1] * b1 + uc[2] * b2 + uc[3] * b3;
    }
    const double u = s[1] / s[0], v = s[2] / s[0], w = s[3] / s[0];
    const double pr = eos_pressure(ph, s[0], u, v, w, s[4]);
    const size_t n = (size_t)inpoel[4 * (size_t)e + i], N = (size_t)nnode;
    atomicAdd(out + n, s[0]); atomicAdd(out + N + n, u); atomicAdd(out + 2 * N + n, v);
    atomicAdd(out + 3 * N + n, w); atomicAdd(out + 4 * N + n, s[4] / s[0]);
    atomicAdd(out + 5 * N + n, pr); atomicAdd(count + n, 1.0);
  }
}
__global__ __launch_bounds__(256) void k_avg_finish(int nnode, double* __restrict__ out,
                                                    const double* __restrict__ count)
{
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nnode) return;
#pragma unroll
  for (int c = 0; c < 6; ++c) out[(size_t)c * nnode + n] /= count[n];
}

// tet volumes of a mesh that has no device layout yet (mesh-less DGPDE::initialize):
// tk::genGeoElemTet's triple product / 6 (src/Mesh/DerivedData.cpp:1457-1491)
__global__ __launch_bounds__(256) void k_tet_volumes(int nelem, int stride, const int* __restrict__ inpoel,
                                                     const double* __restrict__ x, const double* __restrict__ y,
                                                     const double* __restrict__ z, double* __restrict__ vol)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nelem) return;
  const int A = inpoel[e], B = inpoel[(size_t)stride + e], C = inpoel[(size_t)2 * stride + e],
            D = inpoel[(size_t)3 * stride + e];
  const double ba[3] = { x[B] - x[A], y[B] - y[A], z[B] - z[A] };
  const double ca[3] = { x[C] - x[A], y[C] - y[A], z[C] - z[A] };
  const double da[3] = { x[D] - x[A], y[D] - y[A], z[D] - z[A] };
  const double cx = ca[1] * da[2] - ca[2] * da[1], cy = ca[2] * da[0] - ca[0] * da[2],
               cz = ca[0] * da[1] - ca[1] * da[0];
  vol[e] = (ba[0] * cx + ba[1] * cy + ba[2] * cz) / 6.0;
}

// ================================================================ p-adaptive DG
// DG::eval_ndof (src/Inciter/DG.cpp:1088-1163): a P1 tet stays P1 when the
// physical gradient of any conserved variable exceeds tolref, else becomes P0
__global__ __launch_bounds__(256) void k_eval_ndof(DevMesh m, const double* __restrict__ U,
                                                   double tolref, int* __restrict__ ndofel)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m.nie) return;
  if (ndofel[e] != 4) return;
  ElemGeom g;
  load_geom(m, e, g);
  double ji[3][3];
  inverse_jacobian(g, ji);
  int sign = 0;
  for (int c = 0; c < m.ncomp; ++c) {
    const double* u = U + ((size_t)e * m.ncomp + c) * 4;
    const double d0 = 2 * u[1], d1 = u[1] + 3.0 * u[2], d2 = u[1] + u[2] + 4.0 * u[3];
    const double gx = d0 * ji[0][0] + d1 * ji[1][0] + d2 * ji[2][0];
    const double gy = d0 * ji[0][1] + d1 * ji[1][1] + d2 * ji[2][1];
    const double gz = d0 * ji[0][2] + d1 * ji[1][2] + d2 * ji[2][2];
    if (sqrt(gx * gx + gy * gy + gz * gz) > tolref) ++sign;
  }
  ndofel[e] = sign > 0 ? 4 : 1;
}

// DG::propagate_ndof (DG.cpp:1284-1313), Jacobi: face neighbours of a P1 tet
// become P1.  Ghost entries (rows >= nie) are the owners' values and are copied.
__global__ __launch_bounds__(256) void k_propagate_ndof(DevMesh m, const int* __restrict__ in,
                                                        int* __restrict__ out)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m.ne) return;
  int v = in[e];
  if (e < m.nie && v != 4) {
#pragma unroll
    for (int lf = 0; lf < 4; ++lf) {
      const int nb = m.nbr[(size_t)lf * m.stride + e];
      if (nb >= 0 && in[nb] == 4) v = 4;
    }
  }
  out[e] = v;
}

// DG::solve (DG.cpp:1451-1469): high-order DOFs of P0 tets are zeroed at stage 0
__global__ __launch_bounds__(256) void k_pdg_zero(DevMesh m, const int* __restrict__ ndofel,
                                                  double* __restrict__ U)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m.ne || ndofel[e] != 1) return;
  for (int c = 0; c < m.ncomp; ++c)
#pragma unroll
    for (int k = 1; k < 4; ++k) U[((size_t)e * m.ncomp + c) * 4 + k] = 0.0;
}

// face records in task order: tgeo[slot] = fgeo[task_f[slot]] for the used slots of the padded lists
__global__ __launch_bounds__(256) void k_task_geo(size_t nslot, const int* __restrict__ task_a,
                                                  const int* __restrict__ task_f, const double* __restrict__ fgeo,
                                                  double* __restrict__ tgeo)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nslot) return;
  double g[4] = { 0.0, 0.0, 0.0, 0.0 };
  if (task_a[i] >= 0) load_row<4>(fgeo, task_f[i], g);
  double2* o = reinterpret_cast<double2*>(tgeo + 4 * i);
  o[0] = make_double2(g[0], g[1]); o[1] = make_double2(g[2], g[3]);
}

__global__ void k_fill_int(int* __restrict__ p, int n, int v)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ================================================================ scalar transport
// dg::Transport (src/PDE/Transport/DGTransport.hpp:129-186) for ONE transported
// scalar (BASELINE config 1: slot_cyl, DG-P0, Upwind): rows of NDOF doubles,
// U[e*NDOF + k].  Same mesh layout, face codes and quadrature tables as CompFlow;
// element-centric (every tet visits its 4 faces, R written once).
namespace tr {

// TransportProblemSlotCyl::solution (src/PDE/Transport/Problem/SlotCyl.cpp:30-110), c = 0
__device__ double solution_slot_cyl(double x, double y, double t)
{
  const double T = t, R0 = 0.15, PI = 3.14159265358979323846;
  double s = 0.0;
  double x0 = 0.5, y0 = 0.25;
  double r = sqrt((x0 - 0.5) * (x0 - 0.5) + (y0 - 0.5) * (y0 - 0.5));
  const double kx = 0.5 + r * sin(T), ky = 0.5 - r * cos(T);
  x0 = 0.25; y0 = 0.5;
  r = sqrt((x0 - 0.5) * (x0 - 0.5) + (y0 - 0.5) * (y0 - 0.5));
  const double hx = 0.5 + r * sin(T - PI / 2.0), hy = 0.5 - r * cos(T - PI / 2.0);
  x0 = 0.5; y0 = 0.75;
  r = sqrt((x0 - 0.5) * (x0 - 0.5) + (y0 - 0.5) * (y0 - 0.5));
  const double cx = 0.5 + r * sin(T + PI), cy = 0.5 - r * cos(T + PI);
  const double i1x = 0.525, i1y = cy - r * cos(asin(0.025 / r)), i2x = 0.525, i2y = 0.8,
               i3x = 0.475, i3y = 0.8;
  const double ct = cos(T), st = sin(T);
  const double ri1x = 0.5 + ct * (i1x - 0.5) - st * (i1y - 0.5), ri1y = 0.5 + st * (i1x - 0.5) + ct * (i1y - 0.5);
  const double ri2x = 0.5 + ct * (i2x - 0.5) - st * (i2y - 0.5), ri2y = 0.5 + st * (i2x - 0.5) + ct * (i2y - 0.5);
  const double ri3x = 0.5 + ct * (i3x - 0.5) - st * (i3y - 0.5), ri3y = 0.5 + st * (i3x - 0.5) + ct * (i3y - 0.5);
  const double v1x = ri2x - ri1x, v1y = ri2y - ri1y, v2x = ri3x - ri2x, v2y = ri3y - ri2y;
  const double v1 = sqrt(v1x * v1x + v1y * v1y), v2 = sqrt(v2x * v2x + v2y * v2y);
  r = sqrt((x - kx) * (x - kx) + (y - ky) * (y - ky)) / R0;          // cone
  if (r < 1.0) s = 0.6 * (1.0 - r);
  r = sqrt((x - hx) * (x - hx) + (y - hy) * (y - hy)) / R0;          // hump
  if (r < 1.0) s = 0.2 * (1.0 + cos(PI * fmin(r, 1.0)));
  r = sqrt((x - cx) * (x - cx) + (y - cy) * (y - cy)) / R0;          // slotted cylinder
  const double d1 = (v1x * (y - ri1y) - (x - ri1x) * v1y) / v1;
  const double d2 = (v2x * (y - ri2y) - (x - ri2x) * v2y) / v2;
  if (r < 1.0 && (d1 > 0.05 || d1 < 0.0 || d2 < 0.0)) s = 0.6;
  return s;
}

// problem ids: 5 slot_cyl, 8 cyl_advect (CylAdvect.cpp:28-60), 9 gauss_hump (GaussHump.cpp:28-56)
__device__ __forceinline__ double solution(int problem, double x, double y, double /*z*/, double t)
{
  if (problem == 5) return solution_slot_cyl(x, y, t);
  const double x0 = 0.25 + 0.1 * t, y0 = 0.25 + 0.1 * t;
  const double d2 = (x - x0) * (x - x0) + (y - y0) * (y - y0);
  if (problem == 8) return sqrt(d2) < 0.2 ? 1.0 : 0.0;
  if (problem == 9) return 1.0 * exp(-d2 / (2.0 * 0.005));
  return 0.0;
}

// Problem::prescribedVelocity: SlotCyl.cpp:152-170 solid-body rotation about (0.5, 0.5);
// CylAdvect.cpp:114-129, GaussHump.cpp:110-125 constant (0.1, 0.1, 0)
__device__ __forceinline__ void velocity(int problem, double x, double y, double /*z*/, double* v)
{
  if (problem == 5) { v[0] = 0.5 - y; v[1] = x - 0.5; v[2] = 0.0; }
  else { v[0] = 0.1; v[1] = 0.1; v[2] = 0.0; }
}

// Upwind::flux, src/PDE/Integrate/Riemann/Upwind.hpp:35-55
__device__ __forceinline__ double upwind(const double* fn, double ul, double ur, const double* v)
{
  const double swave = v[0] * fn[0] + v[1] * fn[1] + v[2] * fn[2];
  const double splus = 0.5 * (swave + fabs(swave));
  const double sminus = 0.5 * (swave - fabs(swave));
  return splus * ul + sminus * ur;
}

template <int NDOF> __device__ __forceinline__ void load(const double* __restrict__ U, int e, double* u)
{
#pragma unroll
  for (int k = 0; k < NDOF; ++k) u[k] = U[(size_t)e * NDOF + k];
}
template <int NDOF> __device__ __forceinline__ double state(const double* u, const double* B)
{
  double a = u[0];
#pragma unroll
  for (int k = 1; k < NDOF; ++k) a += u[k] * B[k];
  return a;
}

// BC codes of the nbr plane: 1 Dirichlet, 3 Extrapolate, 4 Inlet, 5 Outlet
// (DGTransport.hpp:163-168, 276-352).  With p-adaptive DG (m.ndofel) a P0 tet
// contributes its mean only, gets no high-order update and no volume term, and
// a face uses NGfa(max of the two sides) points (Surface.cpp:81-86; Boundary.cpp:94):
// the velocity and the Dirichlet state vary along the face, so the point count
// must match the reference's exactly.
template <int NDOF>
__global__ __launch_bounds__(256) void k_rhs(DevMesh m, Phys ph, double t,
                                             const double* __restrict__ U, double* __restrict__ R)
{
  const int e = xcd_tile(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (e >= m.nie) return;
  const Tables<NDOF>& T = tab<NDOF>();
  constexpr int NGF = Tables<NDOF>::NGF, NGV = Tables<NDOF>::NGV;
  const int stride = m.stride;
  const bool pdg = NDOF == 4 && m.ndofel != nullptr;
  const bool p0 = pdg && m.ndofel[e] == 1;
  double acc[NDOF], u[NDOF];
#pragma unroll
  for (int k = 0; k < NDOF; ++k) acc[k] = 0.0;
  load<NDOF>(U, e, u);
  if (p0) {
#pragma unroll
    for (int k = 1; k < NDOF; ++k) u[k] = 0.0;
  }
  ElemGeom g;
  load_geom(m, e, g);
#pragma unroll 1
  for (int lf = 0; lf < 4; ++lf) {
    const int nb = m.nbr[(size_t)lf * stride + e];
    if (nb == -1) continue;                     // boundary face without a BC
    const int info = m.finfo[(size_t)lf * stride + e];
    const int f = m.fid[(size_t)lf * stride + e];
    const double area = m.farea[f];
    const double fn[3] = { m.fnx[f], m.fny[f], m.fnz[f] };
    const bool own_left = (info >> 6) & 1;
    double un[NDOF];
    bool p0n = true;
    if (nb >= 0) {
      load<NDOF>(U, nb, un);
      p0n = pdg && m.ndofel[nb] == 1;
      if (p0n) {
#pragma unroll
        for (int k = 1; k < NDOF; ++k) un[k] = 0.0;
      }
    }
    const bool one = pdg && p0 && p0n;          // NGfa(1) = 1: the face centroid, weight 1
    const int ng = one ? 1 : NGF;
    const int own_code = lpofa(lf, 0) | (lpofa(lf, 1) << 2) | (lpofa(lf, 2) << 4);
#pragma unroll 1
    for (int ig = 0; ig < ng; ++ig) {
      const double s0 = one ? 1.0 / 3.0 : T.fs[ig][0], s1 = one ? 1.0 / 3.0 : T.fs[ig][1],
                   s2 = one ? 1.0 / 3.0 : T.fs[ig][2];
      double Bo[NDOF];
      {
        double xi, eta, zeta;
        nbr_ref_coords(own_code, s0, s1, s2, xi, eta, zeta);
        eval_basis<NDOF>(xi, eta, zeta, Bo);
      }
      const double so = state<NDOF>(u, Bo);
      double P[3], v[3], sn;
      face_point(g, lf, s0, s1, s2, P);
      if (nb >= 0) {
        double xi, eta, zeta, Bn[NDOF];
        nbr_ref_coords(info, s0, s1, s2, xi, eta, zeta);
        eval_basis<NDOF>(xi, eta, zeta, Bn);
        sn = state<NDOF>(un, Bn);
      } else {
        const int bc = -nb - 1;
        sn = (bc == 4) ? 0.0 : (bc == 1) ? solution(ph.problem, P[0], P[1], P[2], t) : so;
      }
      velocity(ph.problem, P[0], P[1], P[2], v);
      const double fl = own_left ? upwind(fn, so, sn, v) : upwind(fn, sn, so, v);
      const double wt = (own_left ? -1.0 : 1.0) * (one ? 1.0 : T.fw[ig]) * area;
      acc[0] += wt * fl;
      if (!p0) {
#pragma unroll
        for (int k = 1; k < NDOF; ++k) acc[k] += wt * fl * Bo[k];
      }
    }
  }
  if constexpr (NDOF > 1) {        // volInt, src/PDE/Integrate/Volume.cpp:20-168
    if (!p0) {
      const double vol = m.vol[e];
      double ji[3][3];
      inverse_jacobian(g, ji);
#pragma unroll 1
      for (int ig = 0; ig < NGV; ++ig) {
        const double xi = T.vc[ig][0], eta = T.vc[ig][1], zeta = T.vc[ig][2];
        const double w0 = 1.0 - xi - eta - zeta;
        double P[3], v[3];
#pragma unroll
        for (int d = 0; d < 3; ++d)
          P[d] = g.p[0][d] * w0 + g.p[1][d] * xi + g.p[2][d] * eta + g.p[3][d] * zeta;
        const double sc = state<NDOF>(u, T.vB[ig]);
        velocity(ph.problem, P[0], P[1], P[2], v);
        const double wt = T.vw[ig] * vol;
#pragma unroll
        for (int k = 1; k < NDOF; ++k) {
          const double g0 = T.vdB[ig][0][k], g1 = T.vdB[ig][1][k], g2 = T.vdB[ig][2][k];
          const double dx = g0 * ji[0][0] + g1 * ji[1][0] + g2 * ji[2][0];
          const double dy = g0 * ji[0][1] + g1 * ji[1][1] + g2 * ji[2][1];
          const double dz = g0 * ji[0][2] + g1 * ji[1][2] + g2 * ji[2][2];
          acc[k] += wt * (v[0] * sc * dx + v[1] * sc * dy + v[2] * sc * dz);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NDOF; ++k) R[(size_t)e * NDOF + k] = acc[k];
}

// Superbee_P1 (src/PDE/Limiter.cpp:155-316) for one scalar; in place (only
// neighbour means are read and a mean never changes)
template <int NDOF>
__global__ __launch_bounds__(256) void k_superbee(DevMesh m, double* __restrict__ U)
{
  const int e = xcd_tile(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (e >= m.nie) return;
  if constexpr (NDOF > 1) {
    if (m.ndofel && m.ndofel[e] == 1) return;        // Limiter.cpp:179-180
    const Tables<NDOF>& T = tab<NDOF>();
    constexpr int NGF = Tables<NDOF>::NGF;
    double u[NDOF];
    load<NDOF>(U, e, u);
    double uMin = u[0], uMax = u[0], phi = 1.0;
#pragma unroll
    for (int lf = 0; lf < 4; ++lf) {
      const int nb = m.nbr[(size_t)lf * m.stride + e];
      if (nb < 0) continue;
      const double v = U[(size_t)nb * NDOF];
      uMin = fmin(uMin, v); uMax = fmax(uMax, v);
    }
#pragma unroll 1
    for (int lf = 0; lf < 4; ++lf)
#pragma unroll
      for (int ig = 0; ig < NGF; ++ig) {
        const double uNeg = state<NDOF>(u, T.fB[lf][ig]) - u[0];
        double pg;
        if (uNeg > 1.0e-14)       pg = fmin(1.0, (uMax - u[0]) / (2.0 * uNeg));
        else if (uNeg < -1.0e-14) pg = fmin(1.0, (uMin - u[0]) / (2.0 * uNeg));
        else                      pg = 1.0;
        pg = fmax(0.0, fmax(fmin(2.0 * pg, 1.0), fmin(pg, 2.0)));
        phi = fmin(phi, pg);
      }
#pragma unroll
    for (int k = 1; k < 4; ++k) U[(size_t)e * NDOF + k] = phi * u[k];
  }
}

// WENO_P1 (src/PDE/Limiter.cpp:29-153) for one scalar: Jacobi, Uin -> Uout (all rows)
template <int NDOF>
__global__ __launch_bounds__(256) void k_weno(DevMesh m, double cweight, const double* __restrict__ Uin,
                                              double* __restrict__ Uout)
{
  const int e = xcd_tile(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (e >= m.ne) return;
  double r[NDOF];
  load<NDOF>(Uin, e, r);
  if constexpr (NDOF > 1) {
    if (e < m.nie) {
      double g[5][3], wd[5], wtot = 0.0;
      int nb[4];
#pragma unroll
      for (int lf = 0; lf < 4; ++lf) nb[lf] = m.nbr[(size_t)lf * m.stride + e];
#pragma unroll
      for (int d = 0; d < 3; ++d) g[0][d] = r[1 + d];
#pragma unroll
      for (int is = 1; is < 5; ++is)
#pragma unroll
        for (int d = 0; d < 3; ++d)
          g[is][d] = (nb[is - 1] >= 0) ? Uin[(size_t)nb[is - 1] * NDOF + 1 + d] : 0.0;
#pragma unroll
      for (int is = 0; is < 5; ++is) {
        const double wst = (is == 0) ? cweight : (nb[is - (is > 0)] >= 0 ? 1.0 : 0.0);
        const double osc = sqrt(g[is][0] * g[is][0] + g[is][1] * g[is][1] + g[is][2] * g[is][2]);
        const double q = 1.0e-8 + osc;
        wd[is] = wst / (q * q);
        wtot += wd[is];
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        double a = 0.0;
#pragma unroll
        for (int is = 0; is < 5; ++is) a += (wd[is] / wtot) * g[is][d];
        r[1 + d] = a;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NDOF; ++k) Uout[(size_t)e * NDOF + k] = r[k];
}


template <int NDOF>
__global__ __launch_bounds__(256) void k_init(DevMesh m, Phys ph, double t, double* __restrict__ U)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m.nie) return;
  const QuadTet& Q = c_qinit[order_index<NDOF>()];
  ElemGeom g;
  load_geom(m, e, g);
  const double vol = m.vol[e];
  double acc[NDOF];
#pragma unroll
  for (int k = 0; k < NDOF; ++k) acc[k] = 0.0;
#pragma unroll 1
  for (int ig = 0; ig < Q.ng; ++ig) {
    const double xi = Q.c[ig][0], eta = Q.c[ig][1], zeta = Q.c[ig][2];
    const double w0 = 1.0 - xi - eta - zeta;
    double P[3], B[NDOF];
#pragma unroll
    for (int d = 0; d < 3; ++d)
      P[d] = g.p[0][d] * w0 + g.p[1][d] * xi + g.p[2][d] * eta + g.p[3][d] * zeta;
    eval_basis<NDOF>(xi, eta, zeta, B);
    const double sv = solution(ph.problem, P[0], P[1], P[2], t);
    const double wt = Q.w[ig] * vol;
    acc[0] += wt * sv;
#pragma unroll
    for (int k = 1; k < NDOF; ++k) acc[k] += wt * sv * B[k];
  }
  const double f[10] = { vol, vol / 10.0, vol * 3.0 / 10.0, vol * 3.0 / 5.0, vol / 35.0,
                         vol / 21.0, vol / 14.0, vol / 7.0, vol * 3.0 / 14.0, vol * 3.0 / 7.0 };
#pragma unroll
  for (int k = 0; k < NDOF; ++k) U[(size_t)e * NDOF + k] = acc[k] / f[k];
}

template <int NDOF>
__global__ __launch_bounds__(256) void k_diag(DevMesh m, Phys ph, double t_new,
                                              const double* __restrict__ U, double* __restrict__ part)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  double v[15];
#pragma unroll
  for (int i = 0; i < 15; ++i) v[i] = 0.0;
  if (e < m.nie) {
    const bool p0 = NDOF > 1 && m.ndofel && m.ndofel[e] == 1;   // ElemDiagnostics.cpp:144
    const QuadTet& Q = p0 ? c_qdiag[0] : c_qdiag[order_index<NDOF>()];
    ElemGeom g;
    load_geom(m, e, g);
    const double vol = m.vol[e];
    double u[NDOF];
    load<NDOF>(U, e, u);
    if (p0) {
#pragma unroll
      for (int k = 1; k < NDOF; ++k) u[k] = 0.0;
    }
#pragma unroll 1
    for (int ig = 0; ig < Q.ng; ++ig) {
      const double xi = Q.c[ig][0], eta = Q.c[ig][1], zeta = Q.c[ig][2];
      const double w0 = 1.0 - xi - eta - zeta;
      double P[3], B[NDOF];
#pragma unroll
      for (int d = 0; d < 3; ++d)
        P[d] = g.p[0][d] * w0 + g.p[1][d] * xi + g.p[2][d] * eta + g.p[3][d] * zeta;
      eval_basis<NDOF>(xi, eta, zeta, B);
      const double uu = state<NDOF>(u, B);
      const double d = uu - solution(ph.problem, P[0], P[1], P[2], t_new);
      const double wt = Q.w[ig] * vol;
      v[0] += wt * uu * uu;
      v[5] += wt * d * d;
      v[10] = fmax(v[10], fabs(d));
    }
  }
  diag_block_reduce(v, part);
}

template <int NDOF>
__global__ void k_mass(DevMesh m, double* __restrict__ L)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m.ne) return;
  const double vol = m.vol[e];
  const double f[10] = { vol, vol / 10.0, vol * 3.0 / 10.0, vol * 3.0 / 5.0, vol / 35.0,
                         vol / 21.0, vol / 14.0, vol / 7.0, vol * 3.0 / 14.0, vol * 3.0 / 7.0 };
#pragma unroll
  for (int k = 0; k < NDOF; ++k) L[(size_t)e * NDOF + k] = f[k];
}

template <int NDOF>
__global__ __launch_bounds__(256) void k_rk(DevMesh m, double a, double b, const double* __restrict__ dt,
                                            const double* __restrict__ Un, const double* __restrict__ R,
                                            const double* U, double* Uout)
{
  constexpr double imf[10] = { 1.0, 10.0, 10.0 / 3.0, 5.0 / 3.0, 35.0, 21.0, 14.0, 7.0,
                               14.0 / 3.0, 7.0 / 3.0 };
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)m.nie * NDOF) return;
  const int e = (int)(i / NDOF);
  const int k = (int)(i - (size_t)e * NDOF);
  double f = imf[0];
#pragma unroll
  for (int j = 1; j < NDOF; ++j) f = (k == j) ? imf[j] : f;
  const double dtv = dt[0] / m.vol[e];
  Uout[i] = a * Un[i] + b * (U[i] + dtv * f * R[i]);
}

}  // namespace tr

// ------------------------------------------------- host <-> device rows
// Host rows (caller's element numbering) <-> device rows (device numbering):
// both are element-major, so this is a row permutation; consecutive lanes move
// consecutive doubles of one row.
__global__ __launch_bounds__(256) void k_rows_in(const double* __restrict__ host, int nprop,
                                                 const int* __restrict__ d2h, int n0, int n1,
                                                 double* __restrict__ dev)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)(n1 - n0) * nprop;
  if (i >= n) return;
  const int d = n0 + (int)(i / nprop), p = (int)(i % nprop);
  dev[(size_t)d * nprop + p] = host[(size_t)d2h[d] * nprop + p];
}

__global__ __launch_bounds__(256) void k_rows_out(const double* __restrict__ dev, int nprop,
                                                  const int* __restrict__ d2h, int n0, int n1,
                                                  double* __restrict__ host)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)(n1 - n0) * nprop;
  if (i >= n) return;
  const int d = n0 + (int)(i / nprop), p = (int)(i % nprop);
  host[(size_t)d2h[d] * nprop + p] = dev[(size_t)d * nprop + p];
}

// copy rows [0,n) -- used by the WENO ping-pong
__global__ void k_copy_rows(const double* __restrict__ src, double* __restrict__ dst, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// DG::resizePostAMR (src/Inciter/DG.cpp:1597-1605): a child takes its parent's row.
// to-row d (device order of the new mesh) <- from-row h2d_from[parent[d2h_to[d]]]
__global__ __launch_bounds__(256) void k_state_transfer(int nrow, int nchunk, const int* __restrict__ d2h_to,
                                                        const int* __restrict__ parent,
                                                        const int* __restrict__ h2d_from,
                                                        const double2* __restrict__ Ufrom,
                                                        double2* __restrict__ Uto)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)nrow * nchunk) return;
  const int d = (int)(i / nchunk), p = (int)(i - (size_t)d * nchunk);
  Uto[i] = Ufrom[(size_t)h2d_from[parent[d2h_to[d]]] * nchunk + p];
}
__global__ __launch_bounds__(256) void k_state_transfer1(int nrow, int nprop, const int* __restrict__ d2h_to,
                                                         const int* __restrict__ parent,
                                                         const int* __restrict__ h2d_from,
                                                         const double* __restrict__ Ufrom,
                                                         double* __restrict__ Uto)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)nrow * nprop) return;
  const int d = (int)(i / nprop), p = (int)(i - (size_t)d * nprop);
  Uto[i] = Ufrom[(size_t)h2d_from[parent[d2h_to[d]]] * nprop + p];
}

// ------------------------------------------------------------- halo
// DG::next / DG::lim send side (src/Inciter/DG.cpp:1023-1036, 1266-1279):
// slab row j = U[send_elem[j]] (element-major rows of nprop doubles)
__global__ void k_halo_pack(const double* __restrict__ U, int nprop,
                            const int* __restrict__ send_elem, int nsend,
                            double* __restrict__ slab, const int* __restrict__ ndofel)
{
  // with p-adaptive DG the tet's ndof travels as one more column of its row
  // (DG.cpp:1032,1275: ndof is piggy-backed on comsol / comlim)
  const int w = nprop + (ndofel ? 1 : 0);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)nsend * w) return;
  const int j = (int)(i / w), p = (int)(i - (size_t)j * w);
  const int e = send_elem[j];
  slab[i] = (p < nprop) ? U[(size_t)e * nprop + p] : (double)ndofel[e];
}

// the common case (even row length, no ndof column): 16-byte chunks
__global__ __launch_bounds__(256) void k_halo_pack2(const double2* __restrict__ U, int nchunk,
                                                    const int* __restrict__ send_elem, int nsend,
                                                    double2* __restrict__ slab)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nsend * nchunk) return;
  const int j = i / nchunk, p = i - j * nchunk;
  slab[i] = U[(size_t)send_elem[j] * nchunk + p];
}

// DG::lim / DG::dt receive side (DG.cpp:1239-1247, 1372-1380): ghost rows
// [nie, nie+nrecv) are contiguous, so unpacking is one contiguous copy
__global__ void k_halo_unpack(const double* __restrict__ slab, int nprop, int nie, int nrecv,
                              double* __restrict__ U, int* __restrict__ ndofel)
{
  const int w = nprop + (ndofel ? 1 : 0);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)nrecv * w) return;
  const int j = (int)(i / w), p = (int)(i - (size_t)j * w);
  if (p < nprop) U[(size_t)(nie + j) * nprop + p] = slab[i];
  else ndofel[nie + j] = (int)slab[i];
}

// ================================================================ launchers

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

#ifdef QDG_STAMPS
hipError_t read_stamps(unsigned long long* out16, bool reset)
{
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamp), 16 * sizeof(unsigned long long));
  if (e == hipSuccess && reset) {
    unsigned long long z[16] = { 0 };
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z));
  }
  return e;
}
#endif

hipError_t upload_tables(const Tables<1>& t1, const Tables<4>& t4, const Tables<10>& t10,
                         const QuadTet* qinit, const QuadTet* qdiag)
{
  hipError_t e;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_tab1), &t1, sizeof(t1))) != hipSuccess) return e;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_tab4), &t4, sizeof(t4))) != hipSuccess) return e;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_tab10), &t10, sizeof(t10))) != hipSuccess) return e;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_qinit), qinit, 3 * sizeof(QuadTet))) != hipSuccess) return e;
  if ((e = hipMemcpyToSymbol(HIP_SYMBOL(c_qdiag), qdiag, 3 * sizeof(QuadTet))) != hipSuccess) return e;
  {
    // tables of k_rhs_p2s, from the same rules and basis functions as Tables<10>
    static P2Split ps;
    std::memset(&ps, 0, sizeof(ps));
    for (int m0 = 0; m0 < 4; ++m0) for (int m1 = 0; m1 < 4; ++m1) for (int m2 = 0; m2 < 4; ++m2) {
      if (m0 == m1 || m0 == m2 || m1 == m2) continue;
      const int code = m0 | (m1 << 2) | (m2 << 4), r = perm_rank(code);
      for (int gq = 0; gq < 6; ++gq) {
        double wn[4] = { 0, 0, 0, 0 };
        wn[m0] += t10.fs[gq][0]; wn[m1] += t10.fs[gq][1]; wn[m2] += t10.fs[gq][2];
        double B[10];
        host_basis(10, wn[1], wn[2], wn[3], B);
        for (int k = 0; k < 10; ++k) ps.face[r][gq][k / 5][k % 5] = B[k];
      }
    }
    for (int gq = 0; gq < 6; ++gq) {
      for (int j = 0; j < 3; ++j) ps.fq[gq][j] = t10.fs[gq][j];
      ps.fq[gq][3] = t10.fw[gq];
    }
    for (int gv = 0; gv < 12; ++gv) {
      const int src = gv < 11 ? gv : 0;
      ps.vw[gv] = gv < 11 ? t10.vw[src] : 0.0;
      for (int d = 0; d < 3; ++d) ps.vc[gv][d] = t10.vc[src][d];
      for (int k = 0; k < 10; ++k) {
        ps.vol[gv][k / 5][k % 5] = t10.vB[src][k];
        for (int j = 0; j < 3; ++j) ps.vol[gv][k / 5][5 + 5 * j + k % 5] = t10.vdB[src][j][k];
      }
    }
    if ((e = hipMemcpyToSymbol(HIP_SYMBOL(g_p2s), &ps, sizeof(ps))) != hipSuccess) return e;
  }
  return hipSuccess;
}

// DG-P2 runs through the face-batched kernel (QDG_P2_GENERIC=1 keeps k_rhs<10> for A/B runs)
static bool p2_batched(int ndof)
{
  static const bool generic = std::getenv("QDG_P2_GENERIC") != nullptr;
  return ndof == 10 && !generic;
}

// ... in its two-lanes-per-tet form (QDG_P2_ONE_LANE=1: the one-lane face-batched kernel)
static bool p2_split()
{
  static const bool one = std::getenv("QDG_P2_ONE_LANE") != nullptr;
  return !one;
}

void launch_task_geo(size_t nslot, const int* task_a, const int* task_f, const double* fgeo, double* tgeo,
                     hipStream_t s)
{
  if (nslot == 0) return;
  k_task_geo<<<(unsigned)((nslot + 255) / 256), 256, 0, s>>>(nslot, task_a, task_f, fgeo, tgeo);
}

void launch_rhs(int ndof, const DevMesh& m, const Phys& ph, double t, const double* U, double* R,
                hipStream_t s)
{
  if (m.nie == 0) return;
  if (m.ncomp == 1) {
    QDG_DISPATCH_NDOF(ndof, (tr::k_rhs<N><<<nblk(m.nie, 256), 256, 0, s>>>(m, ph, t, U, R)));
    return;
  }
  if (p2_batched(ndof)) {
    if (p2_split()) {
      QDG_DISPATCH_PROB(ph.problem, (k_rhs_p2s<P, 0><<<nblk(m.nie, 128), 256, 0, s>>>(m, ph, t, U, R, nullptr, 0.0, 0.0, nullptr, nullptr)));
    } else {
      QDG_DISPATCH_PROB(ph.problem, (k_rhs_p2<P, 0><<<nblk(m.nie, 256), 256, 0, s>>>(m, ph, t, U, R, nullptr, 0.0, 0.0, nullptr, nullptr)));
    }
    return;
  }
  QDG_DISPATCH_NDOF(ndof, QDG_DISPATCH_PROB(ph.problem, (k_rhs<N, P, 0><<<nblk(m.nie, 256), 256, 0, s>>>(m, ph, t, U, R, nullptr, 0.0, 0.0, nullptr, nullptr))));
}

// generic RHS with the CFL time step fused in (stage 0): dt = min(vol/delt) * scale, capped to tleft
void launch_rhs_dt(int ndof, const DevMesh& m, const Phys& ph, double t, const double* U, double* R,
                   double* blockmin, double scale, double tleft, double* out_raw, double* out_dt,
                   hipStream_t s)
{
  int nb = nblk(m.nie, 256);
  if (nb == 0) return;
  if (p2_batched(ndof) && p2_split()) {
    nb = nblk(m.nie, 128);
    QDG_DISPATCH_PROB(ph.problem, (k_rhs_p2s<P, 1><<<nb, 256, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr)));
  } else if (p2_batched(ndof)) {
    QDG_DISPATCH_PROB(ph.problem, (k_rhs_p2<P, 1><<<nb, 256, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr)));
  } else {
    QDG_DISPATCH_NDOF(ndof, QDG_DISPATCH_PROB(ph.problem, (k_rhs<N, P, 1><<<nb, 256, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr))));
  }
  k_dt_final<<<1, 256, 0, s>>>(blockmin, nb, scale, tleft, out_raw, out_dt);
}

// generic RHS with the SSP-RK3 update fused in (stages 1, 2): Uout = a*Un + b*(U + dt*R/L)
void launch_rhs_rk(int ndof, const DevMesh& m, const Phys& ph, double t, const double* U, double* Uout,
                   double a, double b, const double* dt, const double* Un, hipStream_t s)
{
  const int nb = nblk(m.nie, 256);
  if (nb == 0) return;
  if (p2_batched(ndof)) {
    if (p2_split()) {
      QDG_DISPATCH_PROB(ph.problem, (k_rhs_p2s<P, 2><<<nblk(m.nie, 128), 256, 0, s>>>(m, ph, t, U, Uout, nullptr, a, b, dt, Un)));
    } else {
      QDG_DISPATCH_PROB(ph.problem, (k_rhs_p2<P, 2><<<nb, 256, 0, s>>>(m, ph, t, U, Uout, nullptr, a, b, dt, Un)));
    }
    return;
  }
  QDG_DISPATCH_NDOF(ndof, QDG_DISPATCH_PROB(ph.problem, (k_rhs<N, P, 2><<<nb, 256, 0, s>>>(m, ph, t, U, Uout, nullptr, a, b, dt, Un))));
}

// P1 fast path; with_dt: also reduce min(vol/delt) into out_raw/out_dt
void launch_rhs_p1(const DevMesh& m, const Phys& ph, double t, const double* U, double* R,
                   bool with_dt, double* blockmin, double scale, double tleft, double* out_raw,
                   double* out_dt, hipStream_t s)
{
  const int nb = nblk(m.nie, 256);
  if (nb == 0) return;
  if (with_dt) {
    QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1<true, false, P><<<nb, 256, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr)));
    k_dt_final<<<1, 256, 0, s>>>(blockmin, nb, scale, tleft, out_raw, out_dt);
  } else {
    QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1<false, false, P><<<nb, 256, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr)));
  }
}

// version 2 of the tile kernel unless the run is p-adaptive (QDG_TILE_V1=1 keeps version 1
// for A/B runs)
static bool tile_v2(const DevMesh& m)
{
  return !m.ndofel && std::getenv("QDG_TILE_V1") == nullptr;
}

// tile / face-task form of the P1 RHS; tiles [first, first+count) (count < 0: all).
// With with_dt the launch that ends at the last tile also reduces the per-tile
// minima to the time step.
void launch_rhs_p1t(const DevMesh& m0, const Phys& ph, double t, const double* U, double* R,
                    bool with_dt, double* blockmin, double scale, double tleft, double* out_raw,
                    double* out_dt, hipStream_t s, int first, int count)
{
  if (m0.ntile == 0) return;
  DevMesh m = m0;
  m.blk0 = first;
  const int nb = count < 0 ? m.ntile - first : count;
  if (nb > 0 && tile_v2(m)) {
    if (with_dt) {
      QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1v<true, false, P><<<nb, TILE_BS, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr)));
    } else {
      QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1v<false, false, P><<<nb, TILE_BS, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr)));
    }
  } else if (nb > 0) {
    if (with_dt) {
      QDG_DISPATCH_PDG(m, QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1t<true, false, P, G><<<nb, TILE_BS, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr))));
    } else {
      QDG_DISPATCH_PDG(m, QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1t<false, false, P, G><<<nb, TILE_BS, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr))));
    }
  }
  if (with_dt && first + nb == m.ntile)
    k_dt_final<<<1, 256, 0, s>>>(blockmin, m.ntile, scale, tleft, out_raw, out_dt);
}

void launch_rhs_p1t_rk(const DevMesh& m0, const Phys& ph, double t, const double* U, double* Uout,
                       double a, double b, const double* dt, const double* Un, hipStream_t s,
                       int first, int count)
{
  if (m0.ntile == 0) return;
  DevMesh m = m0;
  m.blk0 = first;
  const int nb = count < 0 ? m.ntile - first : count;
  if (nb <= 0) return;
  if (tile_v2(m)) {
    QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1v<false, true, P><<<nb, TILE_BS, 0, s>>>(m, ph, t, U, Uout, nullptr, a, b, dt, Un)));
    return;
  }
  QDG_DISPATCH_PDG(m, QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1t<false, true, P, G><<<nb, TILE_BS, 0, s>>>(m, ph, t, U, Uout, nullptr, a, b, dt, Un))));
}

// P1 RHS with the SSP-RK3 update fused in: Uout = a*Un + b*(U + dt*R/L)
void launch_rhs_p1_rk(const DevMesh& m, const Phys& ph, double t, const double* U, double* Uout,
                      double a, double b, const double* dt, const double* Un, hipStream_t s)
{
  const int nb = nblk(m.nie, 256);
  if (nb == 0) return;
  QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1<false, true, P><<<nb, 256, 0, s>>>(m, ph, t, U, Uout, nullptr, a, b, dt, Un)));
}

// 256-row blocks [first, first+count) (count < 0: all)
void launch_superbee(int ndof, const DevMesh& m0, double* U, hipStream_t s, int first, int count)
{
  if (m0.nie == 0 || ndof == 1) return;
  if (m0.ncomp == 1) {
    if (first == 0) QDG_DISPATCH_NDOF(ndof, (tr::k_superbee<N><<<nblk(m0.nie, 256), 256, 0, s>>>(m0, U)));
    return;
  }
  DevMesh m = m0;
  m.blk0 = first;
  const int nb = count < 0 ? (int)nblk(m.nie, 256) - first : count;
  if (nb <= 0) return;
  QDG_DISPATCH_NDOF(ndof, (k_superbee<N><<<nb, 256, 0, s>>>(m, U)));
}

void launch_upd_superbee(const DevMesh& m, const double* dt, const double* U0, const double* R,
                         double* Uout, hipStream_t s)
{
  if (m.nie == 0) return;
  k_upd_superbee<4><<<nblk(m.nie, 256), 256, 0, s>>>(m, dt, U0, R, Uout);
}

void launch_halo_pack_upd(const double* U0, const double* R, const double* dt, const double* vol,
                          const int* send_elem, int nsend, double* slab, hipStream_t s)
{
  if (nsend == 0) return;
  const size_t n = (size_t)nsend * NCOMP * 4;
  k_halo_pack_upd<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(U0, R, dt, vol, send_elem, nsend, slab);
}

void launch_weno(int ndof, const DevMesh& m, double cweight, const double* Uin, double* Uout,
                 hipStream_t s)
{
  if (m.ne == 0 || ndof == 1) return;
  if (m.ncomp == 1) {
    QDG_DISPATCH_NDOF(ndof, (tr::k_weno<N><<<nblk(m.ne, 256), 256, 0, s>>>(m, cweight, Uin, Uout)));
    return;
  }
  if (ndof == 10) k_weno<10, 128><<<nblk(m.ne, 128), 128, 0, s>>>(m, cweight, Uin, Uout);
  else k_weno<4, 256><<<nblk(m.ne, 256), 256, 0, s>>>(m, cweight, Uin, Uout);
}

void launch_copy_planes(const double* src, double* dst, int nprop, int n, int /*stride*/, hipStream_t s)
{
  if (n == 0) return;
  const size_t tot = (size_t)n * nprop;
  k_copy_rows<<<(unsigned)((tot + 255) / 256), 256, 0, s>>>(src, dst, tot);
}

int dt_blocks(const DevMesh& m) { return nblk(m.nie, 256); }

void launch_dt(int ndof, const DevMesh& m, const Phys& ph, const double* U, double* blockmin,
               double scale, double tleft, double* out_raw, double* out_dt, hipStream_t s)
{
  const int nb = dt_blocks(m);
  if (nb > 0)
    QDG_DISPATCH_NDOF(ndof, (k_dt<N><<<nb, 256, 0, s>>>(m, ph, U, blockmin)));
  k_dt_final<<<1, 256, 0, s>>>(blockmin, nb, scale, tleft, out_raw, out_dt);
}

void launch_rk(int ndof, const DevMesh& m, double a, double b, const double* dt, const double* Un,
               const double* R, const double* U, double* Uout, hipStream_t s)
{
  if (m.nie == 0) return;
  if (m.ncomp == 1) {
    QDG_DISPATCH_NDOF(ndof, (tr::k_rk<N><<<(unsigned)(((size_t)m.nie * N + 255) / 256), 256, 0, s>>>(m, a, b, dt, Un, R, U, Uout)));
    return;
  }
  QDG_DISPATCH_NDOF(ndof, (k_rk<N><<<(unsigned)(((size_t)m.nie * NCOMP * N + 255) / 256), 256, 0, s>>>(m, a, b, dt, Un, R, U, Uout)));
}

void launch_mass(int ndof, const DevMesh& m, double* L, hipStream_t s)
{
  if (m.ne == 0) return;
  if (m.ncomp == 1) {
    QDG_DISPATCH_NDOF(ndof, (tr::k_mass<N><<<nblk(m.ne, 256), 256, 0, s>>>(m, L)));
    return;
  }
  QDG_DISPATCH_NDOF(ndof, (k_mass<N><<<nblk(m.ne, 256), 256, 0, s>>>(m, L)));
}

void launch_init(int ndof, const DevMesh& m, const Phys& ph, double t, double* U, hipStream_t s)
{
  if (m.nie == 0) return;
  if (m.ncomp == 1) {
    QDG_DISPATCH_NDOF(ndof, (tr::k_init<N><<<nblk(m.nie, 256), 256, 0, s>>>(m, ph, t, U)));
    return;
  }
  QDG_DISPATCH_NDOF(ndof, QDG_DISPATCH_PROB(ph.problem, (k_init<N, P><<<nblk(m.nie, 256), 256, 0, s>>>(m, ph, t, U))));
}

void launch_diag(int ndof, const DevMesh& m, const Phys& ph, double t_new, const double* U,
                 double* part, double* out, hipStream_t s)
{
  const int nb = nblk(m.nie, 256);
  if (nb > 0 && m.ncomp == 1) {
    QDG_DISPATCH_NDOF(ndof, (tr::k_diag<N><<<nb, 256, 0, s>>>(m, ph, t_new, U, part)));
  } else if (nb > 0)
    QDG_DISPATCH_NDOF(ndof, QDG_DISPATCH_PROB(ph.problem, (k_diag<N, P><<<nb, 256, 0, s>>>(m, ph, t_new, U, part))));
  k_diag_final<<<1, 64, 0, s>>>(part, nb, out);
}

void launch_aos2soa(const double* aos, int nprop, const int* d2h, int n0, int n1, int /*stride*/,
                    double* soa, hipStream_t s)
{
  if (n1 <= n0) return;
  const size_t n = (size_t)(n1 - n0) * nprop;
  k_rows_in<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(aos, nprop, d2h, n0, n1, soa);
}

void launch_soa2aos(const double* soa, int nprop, const int* d2h, int n0, int n1, int /*stride*/,
                    double* aos, hipStream_t s)
{
  if (n1 <= n0) return;
  const size_t n = (size_t)(n1 - n0) * nprop;
  k_rows_out<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(soa, nprop, d2h, n0, n1, aos);
}

void launch_state_transfer(int nrow, int nprop, const int* d2h_to, const int* parent, const int* h2d_from,
                           const double* Ufrom, double* Uto, hipStream_t s)
{
  if (nrow == 0) return;
  if (nprop % 2 == 0) {
    const size_t n = (size_t)nrow * (nprop / 2);
    k_state_transfer<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(nrow, nprop / 2, d2h_to, parent, h2d_from,
                                                                 reinterpret_cast<const double2*>(Ufrom),
                                                                 reinterpret_cast<double2*>(Uto));
  } else {
    const size_t n = (size_t)nrow * nprop;
    k_state_transfer1<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(nrow, nprop, d2h_to, parent, h2d_from, Ufrom, Uto);
  }
}

void launch_halo_pack(const double* U, int nprop, int /*stride*/, const int* send_elem, int nsend,
                      double* slab, hipStream_t s, const int* ndofel)
{
  if (nsend == 0) return;
  if (!ndofel && nprop % 2 == 0) {
    const int nchunk = nprop / 2;
    const size_t n2 = (size_t)nsend * nchunk;
    k_halo_pack2<<<(unsigned)((n2 + 255) / 256), 256, 0, s>>>(reinterpret_cast<const double2*>(U), nchunk,
                                                              send_elem, nsend, reinterpret_cast<double2*>(slab));
    return;
  }
  const size_t n = (size_t)nsend * (nprop + (ndofel ? 1 : 0));
  k_halo_pack<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(U, nprop, send_elem, nsend, slab, ndofel);
}

void launch_halo_unpack(const double* slab, int nprop, int /*stride*/, int nie, int nrecv, double* U,
                        hipStream_t s, int* ndofel)
{
  if (nrecv == 0) return;
  const size_t n = (size_t)nrecv * (nprop + (ndofel ? 1 : 0));
  k_halo_unpack<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(slab, nprop, nie, nrecv, U, ndofel);
}

__global__ __launch_bounds__(256) void k_tr_solution(int problem, int n, const double* __restrict__ x,
                                                     const double* __restrict__ y,
                                                     const double* __restrict__ z, double t,
                                                     double* __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = tr::solution(problem, x[i], y[i], z[i], t);
}

void launch_solution(int ncomp, const Phys& ph, int n, const double* x, const double* y, const double* z,
                     double t, double* out, hipStream_t s)
{
  if (n == 0) return;
  if (ncomp == 1) { k_tr_solution<<<nblk(n, 256), 256, 0, s>>>(ph.problem, n, x, y, z, t, out); return; }
  QDG_DISPATCH_PROB(ph.problem, (k_solution<P><<<nblk(n, 256), 256, 0, s>>>(ph, n, x, y, z, t, out)));
}

// dg::Transport::fieldOutput, src/PDE/Transport/DGTransport.hpp:248-279 (one scalar):
// mean, Problem::solution at the centroid, (analytic - numerical)^2 * vol
__global__ __launch_bounds__(256) void k_tr_field_output(DevMesh m, int problem, int ndof, double t,
                                                         const double* __restrict__ U,
                                                         const double* __restrict__ geoElem, int nrows,
                                                         double* __restrict__ out)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nrows) return;
  size_t h;
  double vol, x, y, z;
  if (geoElem) {
    h = (size_t)e;
    vol = geoElem[4 * h]; x = geoElem[4 * h + 1]; y = geoElem[4 * h + 2]; z = geoElem[4 * h + 3];
  } else {
    h = (size_t)m.d2h[e];
    vol = m.vol[e];
    double q[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = m.inpoel[(size_t)i * m.stride + e];
      q[i][0] = m.x[n]; q[i][1] = m.y[n]; q[i][2] = m.z[n];
    }
    x = (q[0][0] + q[1][0] + q[2][0] + q[3][0]) / 4.0;
    y = (q[0][1] + q[1][1] + q[2][1] + q[3][1]) / 4.0;
    z = (q[0][2] + q[1][2] + q[2][2] + q[3][2]) / 4.0;
  }
  const double u = U[(size_t)e * ndof], sa = tr::solution(problem, x, y, z, t);
  out[h] = u; out[(size_t)nrows + h] = sa; out[2 * (size_t)nrows + h] = (sa - u) * (sa - u) * vol;
}

int field_count(int ncomp, int problem)
{
  if (ncomp == 1) return 3;
  return problem == 3 ? 12 : problem == 4 ? 15 : problem == 7 ? 14 : problem == 10 ? 18 : problem == 0 ? 7 : 6;
}

// geoElem == nullptr: resident state in device rows, output permuted to the caller's
// numbering; else U and geoElem are in the caller's numbering (nrows rows)
void launch_field_output(int ndof, const DevMesh& m, const Phys& ph, double t, const double* U,
                         const double* geoElem, int nrows, double* out, hipStream_t s)
{
  if (nrows == 0) return;
  if (m.ncomp == 1) {
    k_tr_field_output<<<nblk(nrows, 256), 256, 0, s>>>(m, ph.problem, ndof, t, U, geoElem, nrows, out);
  } else {
    QDG_DISPATCH_PROB(ph.problem, (k_field_output<P><<<nblk(nrows, 256), 256, 0, s>>>(m, ph, ndof, t, U, geoElem, nrows, out)));
  }
  if (!geoElem && m.ndofel)
    k_field_ndof<<<nblk(nrows, 256), 256, 0, s>>>(m, nrows, out + (size_t)field_count(m.ncomp, ph.problem) * nrows);
}

void launch_avg_elem_to_node(const Phys& ph, int rdof, int nelem, int nnode, const int* inpoel,
                             const double* U, double* out, double* count, hipStream_t s)
{
  if (nelem > 0) k_avg_elem_to_node<<<nblk(nelem, 256), 256, 0, s>>>(ph, rdof, nelem, nnode, inpoel, U, out, count);
  if (nnode > 0) k_avg_finish<<<nblk(nnode, 256), 256, 0, s>>>(nnode, out, count);
}

void launch_tet_volumes(int nelem, int stride, const int* inpoel, const double* x, const double* y,
                        const double* z, double* vol, hipStream_t s)
{
  if (nelem > 0) k_tet_volumes<<<nblk(nelem, 256), 256, 0, s>>>(nelem, stride, inpoel, x, y, z, vol);
}

// p-adaptive DG: eval_ndof + propagate_ndof + zeroing (stage 0); ndofel/tmp are [ne] ints
void launch_pdg_eval(const DevMesh& m, const double* U, double tolref, int* ndofel, hipStream_t s)
{
  if (m.nie == 0) return;
  k_eval_ndof<<<nblk(m.nie, 256), 256, 0, s>>>(m, U, tolref, ndofel);
}
void launch_pdg_propagate(const DevMesh& m, const int* in, int* out, hipStream_t s)
{
  if (m.ne == 0) return;
  k_propagate_ndof<<<nblk(m.ne, 256), 256, 0, s>>>(m, in, out);
}
void launch_pdg_zero(const DevMesh& m, const int* ndofel, double* U, hipStream_t s)
{
  if (m.ne == 0) return;
  k_pdg_zero<<<nblk(m.ne, 256), 256, 0, s>>>(m, ndofel, U);
}
void launch_fill_int(int* p, int n, int v, hipStream_t s)
{
  if (n > 0) k_fill_int<<<nblk(n, 256), 256, 0, s>>>(p, n, v);
}

}  // namespace qdg
