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
#include "qdg_devfn.hpp"

namespace qdg {

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

// ------------------------------------------------------------- limiters
// Superbee_P1, src/PDE/Limiter.cpp:155-316: only neighbour MEANS are read, so
// the in-place update is order independent.
// Limiter.cpp:283-301: phi = min over the face points of f(phi_gp),
// phi_gp = min(1, (uMax|uMin - u0) / (2 uNeg)), f(p) = max(0, max(min(2p,1), min(p,2))).
// f is non-decreasing, so phi = f(min phi_gp).  All points with uNeg > 0 share the numerator
// uMax - u0 and all points with uNeg < 0 share u0 - uMin, so the smallest phi_gp of either sign
// sits at the LARGEST excursion of that sign: the point loop only tracks the largest positive and
// the most negative uNeg (fmax / fmin drop a NaN excursion, as the reference's two comparisons do),
// and the two candidates are compared by cross-multiplication (a, b >= 0) and divided once per
// component (|uNeg| <= 1e-14 -> phi_gp = 1: not a candidate).  uNeg itself is formed as the
// reference forms it, state - mean, cancellation included.
// one component at a time (short live ranges: the kernel's occupancy is bounded by its registers)
template <int NDOF>
__device__ __forceinline__ double superbee_phi1(const Tables<NDOF>& T, const double (&u)[NDOF], double uMin, double uMax)
{
  constexpr int NGF = Tables<NDOF>::NGF;
  double hi = 0.0, lo = 0.0;
#pragma unroll
  for (int lf = 0; lf < 4; ++lf)
#pragma unroll
    for (int ig = 0; ig < NGF; ++ig) {
      const double* B = T.fB[lf][ig];
      double a = u[0];
#pragma unroll
      for (int k = 1; k < NDOF; ++k) a += u[k] * B[k];        // (state_from's order)
      const double uNeg = a - u[0];
      hi = fmax(hi, uNeg);
      lo = fmin(lo, uNeg);
    }
  const double ap = uMax - u[0], bp = 2.0 * hi;
  const double an = u[0] - uMin, bn = -2.0 * lo;
  double ra = 1.0, rb = 1.0;
  const bool takep = (hi > 1.0e-14) && (ap * rb < ra * bp);
  ra = takep ? ap : ra;
  rb = takep ? bp : rb;
  const bool taken = (lo < -1.0e-14) && (an * rb < ra * bn);
  ra = taken ? an : ra;
  rb = taken ? bn : rb;
  const double pg = fmin(1.0, ra * fast_rcp(rb));
  return fmax(0.0, fmax(fmin(2.0 * pg, 1.0), fmin(pg, 2.0)));
}

// Rows of a 256-row tile back to HBM in two coalesced passes of 128 rows through a 128-row LDS staging
// area (half the LDS of tile_store_rows: the limiter kernels' occupancy).  All 256 threads call; the caller
// guarantees that nobody still reads `lds` (a barrier has been passed since the last read).
// The limited rows are written once and read by the next kernel only after this one has streamed gigabytes
// through the L2: non-temporal stores keep them from displacing the rows of U0 / R that the out-of-tile
// neighbour gathers of the same launch find there (k_upd_superbee 1116 -> 1096 us at 10.1 M tets, 127 -> 123 us
// at 1 M).
#define QDG_ROW_STORE(p, v) qdg::store_nt((p), (v))
template <int NPROP>
__device__ __forceinline__ void tile_store_rows_halves(double* __restrict__ U, int tile_e0, int nrows,
                                                       double* __restrict__ lds, const double* r)
{
  const int tid = threadIdx.x;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();                    // the first half has left the staging area
    if ((tid >> 7) == half) {
      double2* row = reinterpret_cast<double2*>(lds + (size_t)(tid & 127) * NPROP);
#pragma unroll
      for (int j = 0; j < NPROP / 2; ++j) row[j] = make_double2(r[2 * j], r[2 * j + 1]);
    }
    __syncthreads();
    const int base = tile_e0 + half * 128;
    const int left = nrows - base;
    const int nvalid = (left < 0 ? 0 : (left < 128 ? left : 128)) * (NPROP / 2);
    const double2* src = reinterpret_cast<const double2*>(lds);
    double2* dst = reinterpret_cast<double2*>(U + (size_t)base * NPROP);
#pragma unroll
    for (int j = 0; j < NPROP / 4; ++j) {
      const int i = j * 256 + tid;
      if (i < nvalid) QDG_ROW_STORE(dst + i, src[i]);
    }
    if constexpr ((NPROP / 2) % 2 != 0) {         // 128 * NPROP/2 chunks over 256 threads: an odd half pass
      const int i = (NPROP / 4) * 256 + tid;
      if (tid < 128 && i < nvalid) QDG_ROW_STORE(dst + i, src[i]);
    }
  }
}

template <int NDOF>
__global__ __launch_bounds__(256) void k_superbee(DevMesh m, double* __restrict__ U)
{
  if constexpr (NDOF > 1) {
    const Tables<NDOF>& T = tab<NDOF>();
    constexpr int NPROP = NCOMP * NDOF;
    // first the tile's MEANS, [c][row] (10 KB), for the in-tile neighbours; later the staging area of the
    // coalesced row stores (128 rows)
    __shared__ double lds[128 * NPROP];
    // rows [0, nlim): the owned tets, plus the layer-1 ghosts of a chunk with two ghost layers
    const int tile_e0 = (m.blk0 + xcd_tile(blockIdx.x, gridDim.x)) * 256;
    const int e0 = tile_e0 + threadIdx.x;
    const bool active = e0 < m.nlim;
    const int e = active ? e0 : m.nlim - 1;
    const int stride = m.stride;
    int nb[4];
#pragma unroll
    for (int lf = 0; lf < 4; ++lf) nb[lf] = m.nbr[(size_t)lf * stride + e];
    double u[NCOMP][NDOF];
    // the lane reads its own row directly (5.7 TB/s measured for that access, against 4.5 for the
    // detour through LDS: tools/ubench_rowstream.hip); only the MEANS go to LDS
    load_row<NPROP>(U, e, &u[0][0]);
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) lds[c * 256 + threadIdx.x] = u[c][0];
    __syncthreads();
    double uMin[NCOMP], uMax[NCOMP];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) uMin[c] = uMax[c] = u[c][0];
#pragma unroll
    for (int lf = 0; lf < 4; ++lf) {
      if (nb[lf] < 0) continue;
      const int r = nb[lf] - tile_e0;
      // in-tile neighbour: means from LDS (a row beyond the limited range may fall into the id range of
      // a ragged last tile: it always comes from global memory)
      if (nb[lf] < m.nlim && (unsigned)r < 256u) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
          const double v = lds[c * 256 + r];
          uMin[c] = fmin(uMin[c], v);
          uMax[c] = fmax(uMax[c], v);
        }
      } else {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
          const double v = U[fidx(c * NDOF, nb[lf], NPROP)];
          uMin[c] = fmin(uMin[c], v);
          uMax[c] = fmax(uMax[c], v);
        }
      }
    }
    const bool p0row = m.ndofel && m.ndofel[e] == 1;   // pdg: P0 elements are not limited (Limiter.cpp:179-180)
    bool changed = false;
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      const double phi = p0row ? 1.0 : superbee_phi1<NDOF>(T, u[c], uMin[c], uMax[c]);
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const double v = phi * u[c][k];
        changed = changed || (v != u[c][k]);
        u[c][k] = v;
      }
    }
    if (active) halo_fold_row<NPROP>(m, e, &u[0][0]);      // (qdg_step_comm: the comlim pack, folded in)
    // A tile in which the limiter changed no value (phi = 1 or zero slopes on every row: uniform and smooth
    // regions) has nothing to write back -- the limiter works in place -- which halves this pass's traffic there.
    // (The barrier inside is also the one that ends the reads of the means.)
    if (!__syncthreads_or((changed || m.lim_write_all) && active)) return;
    // Out-of-tile neighbours may be read from U while another tile has already
    // stored its limited rows: safe, Superbee never changes a mean.
    tile_store_rows_halves<NPROP>(U, tile_e0, m.nlim, lds, &u[0][0]);
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
  // 32 KB instead of a whole tile of rows (42 KB): four workgroups per CU
  __shared__ double stage[128 * NPROP];    // half a tile of rows: coalesced loads in, coalesced stores out
  __shared__ double mean[NCOMP * 256];     // the tile's U1 means [c][row], for the in-tile neighbours
  __shared__ double sdtv[256];
  const int tid = threadIdx.x;
  const int nown = (m.nie + 255) >> 8;           // tiles of owned rows; the workgroups behind them: layer-1 ghosts
  const int tile = m.blk0 + xcd_tile(blockIdx.x, gridDim.x);
  const int stride = m.stride;
  const double dt = dtp[0];
  if (tile >= nown) {
    // Two ghost layers: the rank limits its layer-1 ghosts [nie, nlim) itself.  Their unlimited U1 rows are in
    // Uout already (received); a neighbour's U1 mean comes from Uout if it is a ghost, else from U0 and R as
    // below (the owned rows of Uout are being written by the other workgroups of this launch: never read here).
    // One lane per row, the row rewritten in place (a few thousand rows: no staging)
    const int g = m.nie + (tile - nown) * 256 + tid;
    if (g >= m.nlim) return;
    double u[NCOMP][NDOF];
    load_row<NPROP>(Uout, g, &u[0][0]);
    double uMin[NCOMP], uMax[NCOMP];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) uMin[c] = uMax[c] = u[c][0];
#pragma unroll
    for (int lf = 0; lf < 4; ++lf) {
      const int nb = m.nbr[(size_t)lf * stride + g];
      if (nb < 0) continue;
      const double dtn = (nb < m.nie) ? dt / m.vol[nb] : 0.0;
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const double v = (nb >= m.nie) ? Uout[fidx(c * NDOF, nb, NPROP)]
                                       : U0[fidx(c * NDOF, nb, NPROP)] + dtn * 1.0 * R[fidx(c * NDOF, nb, NPROP)];
        uMin[c] = fmin(uMin[c], v); uMax[c] = fmax(uMax[c], v);
      }
    }
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      const double phi = superbee_phi1<NDOF>(T, u[c], uMin[c], uMax[c]);
#pragma unroll
      for (int k = 1; k < 4; ++k) u[c][k] = phi * u[c][k];
    }
    store_row<NPROP>(Uout, g, &u[0][0]);
    return;
  }
  const int tile_e0 = tile * 256;
  const int e0 = tile_e0 + tid;
  const bool active = e0 < m.nie;
  const int e = active ? e0 : m.nie - 1;
  sdtv[tid] = dt / m.vol[e];                       // the row's dt / vol, as k_rk forms it
  int nbr[4];
#pragma unroll
  for (int lf = 0; lf < 4; ++lf) nbr[lf] = m.nbr[(size_t)lf * stride + e];
  __syncthreads();
  // the tile's rows of U1 = U0 + dt R / L pass through LDS half a tile at a time: one coalesced pass over
  // both arrays, then every lane of that half takes its row
  double u[NCOMP][NDOF];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();                     // the first half's rows have been taken
    const int base = tile_e0 + half * 128;
    const int left = m.nie - base;
    const int nvalid = (left < 0 ? 0 : (left < 128 ? left : 128)) * NCH;
    const double2* su = reinterpret_cast<const double2*>(U0 + (size_t)base * NPROP);
    const double2* sr = reinterpret_cast<const double2*>(R + (size_t)base * NPROP);
    double2* dst = reinterpret_cast<double2*>(stage);
#pragma unroll
    for (int j = 0; j < NCH / 2; ++j) {
      const int i = j * 256 + tid;
      double2 v = make_double2(1.0, 1.0);
      if (i < nvalid) {
        const double2 a = su[i], b = sr[i];
        const int row = i / NCH, hi = (i - row * NCH) & 1;     // chunk holds modes (0,1) or (2,3)
        const double dtv = sdtv[half * 128 + row];
        const double f0 = hi ? 10.0 / 3.0 : 1.0, f1 = hi ? 5.0 / 3.0 : 10.0;
        v = make_double2(a.x + dtv * f0 * b.x, a.y + dtv * f1 * b.y);
      }
      dst[i] = v;
    }
    __syncthreads();
    if ((tid >> 7) == half) lds_row<NPROP>(stage, tid & 127, &u[0][0]);
  }
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) mean[c * 256 + tid] = u[c][0];
  __syncthreads();
  double uMin[NCOMP], uMax[NCOMP];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) uMin[c] = uMax[c] = u[c][0];
#pragma unroll
  for (int lf = 0; lf < 4; ++lf) {
    const int nb = nbr[lf];
    if (nb < 0) continue;
    const int rr = nb - tile_e0;
    if (nb < m.nie && (unsigned)rr < 256u) {     // (a ghost id may alias the ragged last tile)
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        const double v = mean[c * 256 + rr];
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
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) {
    const double phi = superbee_phi1<NDOF>(T, u[c], uMin[c], uMax[c]);
#pragma unroll
    for (int k = 1; k < 4; ++k) u[c][k] = phi * u[c][k];
  }
  if (active) halo_fold_row<NPROP>(m, e, &u[0][0]);        // (qdg_step_comm: the comlim pack of stage 1, folded in)
  // (every lane has passed the barrier behind the means since it last read the staging area)
  tile_store_rows_halves<NPROP>(Uout, tile_e0, m.nie, stage, &u[0][0]);
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
    // the tile's gradient modes [row][c][3] go to LDS first (the area that later stages the rows): a
    // neighbour inside the tile -- two thirds to three quarters of them in the Morton-ordered numbering --
    // is read from there instead of through scattered global loads (the tiles cover ghost rows too)
#pragma unroll
    for (int c = 0; c < NCOMP; ++c)
#pragma unroll
      for (int d = 0; d < 3; ++d) stage[(threadIdx.x * NCOMP + c) * 3 + d] = r[c][1 + d];
    __syncthreads();
    if (e < m.nlim) {
      const int stride = m.stride;
      const int t0 = blk * BS, tn = (m.ne - t0 < BS) ? m.ne - t0 : BS;    // rows [t0, t0 + tn) are in LDS
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
          if (n >= 0 && (unsigned)(n - t0) < (unsigned)tn) {
            const double* l = stage + ((n - t0) * NCOMP + c) * 3;
            g[is][0] = l[0]; g[is][1] = l[1]; g[is][2] = l[2];
          } else if (n >= 0) {
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
      __syncthreads();                            // the gradient modes in LDS have been read
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
        if (i < nvalid) store_nt(dst + i, src[i]);      // (non-temporal: qdg_devfn.hpp, store_nt)
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
  // eight independent loads in flight per lane (one dependent load per iteration made this
  // one-block kernel 64 us long at 40 770 tiles)
  double v = DBL_MAX;
  int i = threadIdx.x;
  for (; i + 7 * (int)blockDim.x < n; i += 8 * blockDim.x) {
    double w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = blockmin[i + k * blockDim.x];
#pragma unroll
    for (int k = 0; k < 8; ++k) v = fmin(v, w[k]);
  }
  for (; i < n; i += blockDim.x) v = fmin(v, blockmin[i]);
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

// TransportProblemSlotCyl::solution (src/PDE/Transport/Problem/SlotCyl.cpp:30-110); T = t + 2 pi c / ncomp
__device__ double solution_slot_cyl(double x, double y, double T)
{
  const double R0 = 0.15, PI = 3.14159265358979323846;
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

// problem ids: 5 slot_cyl, 8 cyl_advect (CylAdvect.cpp:28-60), 9 gauss_hump (GaussHump.cpp:28-56),
// 11 shear_diff (ShearDiff.cpp:28-68: the analytic solution of the advection-diffusion problem
// with the per-component parameters u0, lambda, diffusivity).  Component c of ncomp.
__device__ __forceinline__ double solution(const Phys& ph, int c, int ncomp, double x, double y, double z, double t)
{
  const int problem = ph.problem;
  if (problem == 5) return solution_slot_cyl(x, y, t + 2.0 * 3.14159265358979323846 / ncomp * c);
  if (problem == 11) {
    const double PI = 3.14159265358979323846;
    const double l0 = ph.sd_lambda[2 * c], l1 = ph.sd_lambda[2 * c + 1];
    const double d0 = ph.sd_diff[3 * c], d1 = ph.sd_diff[3 * c + 1], d2 = ph.sd_diff[3 * c + 2];
    const double phi3s = (l0 * l0 * d1 / d0 + l1 * l1 * d2 / d0) / 12.0;
    const double xs = x - ph.sd_u0[c] * t - 0.5 * (l0 * y + l1 * z) * t;
    return 1.0 / (8.0 * pow(PI, 3.0 / 2.0) * sqrt(d0 * d1 * d2) * pow(t, 3.0 / 2.0) * sqrt(1.0 + phi3s * t * t)) *
           exp(-(xs * xs) / (4.0 * d0 * t * (1.0 + phi3s * t * t)) - y * y / (4.0 * d1 * t) - z * z / (4.0 * d2 * t));
  }
  const double x0 = 0.25 + 0.1 * t, y0 = 0.25 + 0.1 * t;
  const double d2 = (x - x0) * (x - x0) + (y - y0) * (y - y0);
  if (problem == 8) return sqrt(d2) < 0.2 ? 1.0 : 0.0;
  if (problem == 9) return 1.0 * exp(-d2 / (2.0 * 0.005));
  return 0.0;
}

// Problem::prescribedVelocity: SlotCyl.cpp:152-170 solid-body rotation about (0.5, 0.5);
// CylAdvect.cpp:114-129, GaussHump.cpp:110-125 constant (0.1, 0.1, 0); ShearDiff.cpp:140-160
// (u0_c + lambda_2c y + lambda_2c+1 z, 0, 0)
__device__ __forceinline__ void velocity(const Phys& ph, int c, double x, double y, double z, double* v)
{
  if (ph.problem == 5) { v[0] = 0.5 - y; v[1] = x - 0.5; v[2] = 0.0; }
  else if (ph.problem == 11) { v[0] = ph.sd_u0[c] + ph.sd_lambda[2 * c] * y + ph.sd_lambda[2 * c + 1] * z; v[1] = 0.0; v[2] = 0.0; }
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

// component c of a row: rows hold ncomp * NDOF doubles, component-major (mark = c * rdof)
template <int NDOF> __device__ __forceinline__ size_t at(const DevMesh& m, int e, int c)
{
  return ((size_t)e * m.ncomp + c) * NDOF;
}
template <int NDOF> __device__ __forceinline__ void load(const double* __restrict__ U, size_t i0, double* u)
{
#pragma unroll
  for (int k = 0; k < NDOF; ++k) u[k] = U[i0 + k];
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
  const int c = blockIdx.y;                      // one transported scalar per grid row (the scalars do not couple)
  const Tables<NDOF>& T = tab<NDOF>();
  constexpr int NGF = Tables<NDOF>::NGF, NGV = Tables<NDOF>::NGV;
  const int stride = m.stride;
  const bool pdg = NDOF == 4 && m.ndofel != nullptr;
  const bool p0 = pdg && m.ndofel[e] == 1;
  double acc[NDOF], u[NDOF];
#pragma unroll
  for (int k = 0; k < NDOF; ++k) acc[k] = 0.0;
  load<NDOF>(U, at<NDOF>(m, e, c), u);
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
      load<NDOF>(U, at<NDOF>(m, nb, c), un);
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
        sn = (bc == 4) ? 0.0 : (bc == 1) ? solution(ph, c, m.ncomp, P[0], P[1], P[2], t) : so;
      }
      velocity(ph, c, P[0], P[1], P[2], v);
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
        velocity(ph, c, P[0], P[1], P[2], v);
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
  for (int k = 0; k < NDOF; ++k) R[at<NDOF>(m, e, c) + k] = acc[k];
}

// Superbee_P1 (src/PDE/Limiter.cpp:155-316) for one scalar; in place (only
// neighbour means are read and a mean never changes)
template <int NDOF>
__global__ __launch_bounds__(256) void k_superbee(DevMesh m, double* __restrict__ U)
{
  const int e = xcd_tile(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (e >= m.nlim) return;
  if constexpr (NDOF > 1) {
    if (m.ndofel && m.ndofel[e] == 1) return;        // Limiter.cpp:179-180
    const int c = blockIdx.y;
    const Tables<NDOF>& T = tab<NDOF>();
    constexpr int NGF = Tables<NDOF>::NGF;
    double u[NDOF];
    load<NDOF>(U, at<NDOF>(m, e, c), u);
    double uMin = u[0], uMax = u[0], phi = 1.0;
#pragma unroll
    for (int lf = 0; lf < 4; ++lf) {
      const int nb = m.nbr[(size_t)lf * m.stride + e];
      if (nb < 0) continue;
      const double v = U[at<NDOF>(m, nb, c)];
      uMin = fmin(uMin, v); uMax = fmax(uMax, v);
    }
    // (one division per scalar instead of one per face point: superbee_phi1, above)
    phi = superbee_phi1<NDOF>(T, u, uMin, uMax);
#pragma unroll
    for (int k = 1; k < 4; ++k) U[at<NDOF>(m, e, c) + k] = phi * u[k];
  }
}

// WENO_P1 (src/PDE/Limiter.cpp:29-153) for one scalar: Jacobi, Uin -> Uout (all rows)
template <int NDOF>
__global__ __launch_bounds__(256) void k_weno(DevMesh m, double cweight, const double* __restrict__ Uin,
                                              double* __restrict__ Uout)
{
  const int e = xcd_tile(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (e >= m.ne) return;
  const int c = blockIdx.y;
  double r[NDOF];
  load<NDOF>(Uin, at<NDOF>(m, e, c), r);
  if constexpr (NDOF > 1) {
    if (e < m.nlim) {
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
          g[is][d] = (nb[is - 1] >= 0) ? Uin[at<NDOF>(m, nb[is - 1], c) + 1 + d] : 0.0;
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
  for (int k = 0; k < NDOF; ++k) Uout[at<NDOF>(m, e, c) + k] = r[k];
}


template <int NDOF>
__global__ __launch_bounds__(256) void k_init(DevMesh m, Phys ph, double t, double* __restrict__ U)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m.nie) return;
  const int c = blockIdx.y;
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
    const double sv = solution(ph, c, m.ncomp, P[0], P[1], P[2], t);
    const double wt = Q.w[ig] * vol;
    acc[0] += wt * sv;
#pragma unroll
    for (int k = 1; k < NDOF; ++k) acc[k] += wt * sv * B[k];
  }
  const double f[10] = { vol, vol / 10.0, vol * 3.0 / 10.0, vol * 3.0 / 5.0, vol / 35.0,
                         vol / 21.0, vol / 14.0, vol / 7.0, vol * 3.0 / 14.0, vol * 3.0 / 7.0 };
#pragma unroll
  for (int k = 0; k < NDOF; ++k) U[at<NDOF>(m, e, c) + k] = acc[k] / f[k];
}

template <int NDOF>
__global__ __launch_bounds__(256) void k_diag(DevMesh m, Phys ph, double t_new,
                                              const double* __restrict__ U, double* __restrict__ part)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;                      // <= 5 scalars: the diagnostics vector has 5 slots per kind
  double v[15];
#pragma unroll
  for (int i = 0; i < 15; ++i) v[i] = 0.0;
  double l2 = 0.0, l2e = 0.0, lie = 0.0;
  if (e < m.nie) {
    const bool p0 = NDOF > 1 && m.ndofel && m.ndofel[e] == 1;   // ElemDiagnostics.cpp:144
    const QuadTet& Q = p0 ? c_qdiag[0] : c_qdiag[order_index<NDOF>()];
    ElemGeom g;
    load_geom(m, e, g);
    const double vol = m.vol[e];
    double u[NDOF];
    load<NDOF>(U, at<NDOF>(m, e, c), u);
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
      const double d = uu - solution(ph, c, m.ncomp, P[0], P[1], P[2], t_new);
      const double wt = Q.w[ig] * vol;
      l2 += wt * uu * uu;
      l2e += wt * d * d;
      lie = fmax(lie, fabs(d));
    }
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    v[i] = (i == c) ? l2 : 0.0; v[5 + i] = (i == c) ? l2e : 0.0; v[10 + i] = (i == c) ? lie : 0.0;
  }
  diag_block_reduce(v, part + (size_t)blockIdx.y * gridDim.x * 15);
}

template <int NDOF>
__global__ void k_mass(DevMesh m, double* __restrict__ L)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m.ne) return;
  const double vol = m.vol[e];
  const double f[10] = { vol, vol / 10.0, vol * 3.0 / 10.0, vol * 3.0 / 5.0, vol / 35.0,
                         vol / 21.0, vol / 14.0, vol / 7.0, vol * 3.0 / 14.0, vol * 3.0 / 7.0 };
  for (int c = 0; c < m.ncomp; ++c)
#pragma unroll
    for (int k = 0; k < NDOF; ++k) L[at<NDOF>(m, e, c) + k] = f[k];
}

template <int NDOF>
__global__ __launch_bounds__(256) void k_rk(DevMesh m, double a, double b, const double* __restrict__ dt,
                                            const double* __restrict__ Un, const double* __restrict__ R,
                                            const double* U, double* Uout)
{
  constexpr double imf[10] = { 1.0, 10.0, 10.0 / 3.0, 5.0 / 3.0, 35.0, 21.0, 14.0, 7.0,
                               14.0 / 3.0, 7.0 / 3.0 };
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)m.nie * m.ncomp * NDOF) return;
  const int e = (int)(i / ((size_t)m.ncomp * NDOF));
  const int k = (int)(i % NDOF);
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
  const int c = d2h_to[d];
  const int par = parent ? parent[c] : (c >> 3);     // (no list: uniform 1:8 refinement, child 8 e + k of tet e)
  if (par < 0) return;                         // row not served by this source (state migration)
  Uto[i] = Ufrom[(size_t)h2d_from[par] * nchunk + p];
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
  const int c = d2h_to[d];
  const int par = parent ? parent[c] : (c >> 3);
  if (par < 0) return;
  Uto[i] = Ufrom[(size_t)h2d_from[par] * nprop + p];
}

// rows of the resident state <-> a packed buffer (state migration between ranks): packed row j
// is device row drow[j]
__global__ __launch_bounds__(256) void k_rows_gather(size_t n, int nprop, const int* __restrict__ drow,
                                                     const double* __restrict__ U, double* __restrict__ packed)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * nprop) return;
  const size_t j = i / nprop, p = i - j * nprop;
  packed[i] = U[(size_t)drow[j] * nprop + p];
}
__global__ __launch_bounds__(256) void k_rows_scatter(size_t n, int nprop, const int* __restrict__ drow,
                                                      const double* __restrict__ packed, double* __restrict__ U)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * nprop) return;
  const size_t j = i / nprop, p = i - j * nprop;
  U[(size_t)drow[j] * nprop + p] = packed[i];
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

hipError_t upload_tables_p1(const Tables<1>&, const Tables<4>&, const Tables<10>&, const QuadTet*, const QuadTet*);
hipError_t upload_tables_p2(const Tables<1>&, const Tables<4>&, const Tables<10>&, const QuadTet*, const QuadTet*);

// every kernel translation unit keeps its own copy of the constant tables
hipError_t upload_tables(const Tables<1>& t1, const Tables<4>& t4, const Tables<10>& t10,
                         const QuadTet* qinit, const QuadTet* qdiag)
{
  hipError_t e;
  if ((e = upload_tables_here(t1, t4, t10, qinit, qdiag)) != hipSuccess) return e;
  if ((e = upload_tables_p1(t1, t4, t10, qinit, qdiag)) != hipSuccess) return e;
  return upload_tables_p2(t1, t4, t10, qinit, qdiag);
}

void launch_task_geo(size_t nslot, const int* task_a, const int* task_f, const double* fgeo, double* tgeo,
                     hipStream_t s)
{
  if (nslot == 0) return;
  k_task_geo<<<(unsigned)((nslot + 255) / 256), 256, 0, s>>>(nslot, task_a, task_f, fgeo, tgeo);
}

void launch_dt_final(const double* blockmin, int n, double scale, double tleft, double* out_raw,
                     double* out_dt, hipStream_t s)
{
  k_dt_final<<<1, 256, 0, s>>>(blockmin, n, scale, tleft, out_raw, out_dt);
}

// RHS of the orders without a kernel file of their own: scalar Transport (any order), CompFlow
// DG-P0 (k_rhs<1>); CompFlow DG-P2 goes to qdg_rhs_p2.hip, DG-P1 is launched through
// launch_rhs_p1* (qdg_rhs_p1.hip)
void launch_rhs(int ndof, const DevMesh& m, const Phys& ph, double t, const double* U, double* R,
                hipStream_t s)
{
  if (m.nie == 0) return;
  if (m.pde == 1) {
    QDG_DISPATCH_NDOF(ndof, (tr::k_rhs<N><<<dim3(nblk(m.nie, 256), m.ncomp), 256, 0, s>>>(m, ph, t, U, R)));
    return;
  }
  if (ndof == 10) { launch_rhs_p2(m, ph, t, U, R, 0, nullptr, 0.0, 0.0, nullptr, nullptr, s); return; }
  if (ndof == 4) { launch_rhs_p1(m, ph, t, U, R, false, nullptr, 1.0, DBL_MAX, nullptr, nullptr, s); return; }
  QDG_DISPATCH_PROB(ph.problem, (k_rhs<1, P, 0><<<nblk(m.nie, 256), 256, 0, s>>>(m, ph, t, U, R, nullptr, 0.0, 0.0, nullptr, nullptr)));
}

// ... with the CFL time step fused in (stage 0): dt = min(vol/delt) * scale, capped to tleft
void launch_rhs_dt(int ndof, const DevMesh& m, const Phys& ph, double t, const double* U, double* R,
                   double* blockmin, double scale, double tleft, double* out_raw, double* out_dt,
                   hipStream_t s)
{
  if (ndof == 4) { launch_rhs_p1(m, ph, t, U, R, true, blockmin, scale, tleft, out_raw, out_dt, s); return; }
  int nb = nblk(m.nie, 256);
  if (nb == 0) return;
  if (ndof == 10) {
    nb = p2_rhs_blocks(m);
    launch_rhs_p2(m, ph, t, U, R, 1, blockmin, 0.0, 0.0, nullptr, nullptr, s);
  } else {
    QDG_DISPATCH_PROB(ph.problem, (k_rhs<1, P, 1><<<nb, 256, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr)));
  }
  k_dt_final<<<1, 256, 0, s>>>(blockmin, nb, scale, tleft, out_raw, out_dt);
}

// ... with the SSP-RK3 update fused in (stages 1, 2): Uout = a*Un + b*(U + dt*R/L)
void launch_rhs_rk(int ndof, const DevMesh& m, const Phys& ph, double t, const double* U, double* Uout,
                   double a, double b, const double* dt, const double* Un, hipStream_t s)
{
  const int nb = nblk(m.nie, 256);
  if (nb == 0) return;
  if (ndof == 10) { launch_rhs_p2(m, ph, t, U, Uout, 2, nullptr, a, b, dt, Un, s); return; }
  if (ndof == 4) { launch_rhs_p1_rk(m, ph, t, U, Uout, a, b, dt, Un, s); return; }
  QDG_DISPATCH_PROB(ph.problem, (k_rhs<1, P, 2><<<nb, 256, 0, s>>>(m, ph, t, U, Uout, nullptr, a, b, dt, Un)));
}

// the limiter's row range: [0, nlim) -- the owned tets and, with two ghost layers, the layer-1 ghosts
static DevMesh lim_range(const DevMesh& m0)
{
  DevMesh m = m0;
  if (m.nlim < m.nie) m.nlim = m.nie;
  return m;
}

// 256-row blocks [first, first+count) (count < 0: all)
void launch_superbee(int ndof, const DevMesh& m0, double* U, hipStream_t s, int first, int count)
{
  if (m0.nie == 0 || ndof == 1) return;
  DevMesh m = lim_range(m0);
  if (m.pde == 1) {
    if (first == 0) QDG_DISPATCH_NDOF(ndof, (tr::k_superbee<N><<<dim3(nblk(m.nlim, 256), m.ncomp), 256, 0, s>>>(m, U)));
    return;
  }
  m.blk0 = first;
  const int nb = count < 0 ? (int)nblk(m.nlim, 256) - first : count;
  if (nb <= 0) return;
  QDG_DISPATCH_NDOF(ndof, (k_superbee<N><<<nb, 256, 0, s>>>(m, U)));
}

void launch_upd_superbee(const DevMesh& m, const double* dt, const double* U0, const double* R,
                         double* Uout, hipStream_t s)
{
  if (m.nie == 0) return;
  // (+ one workgroup per 256 layer-1 ghosts of a chunk with two ghost layers: limited in the same launch)
  const int nghost1 = m.nlim > m.nie ? m.nlim - m.nie : 0;
  k_upd_superbee<4><<<nblk(m.nie, 256) + nblk(nghost1, 256), 256, 0, s>>>(m, dt, U0, R, Uout);
}

void launch_halo_pack_upd(const double* U0, const double* R, const double* dt, const double* vol,
                          const int* send_elem, int nsend, double* slab, hipStream_t s)
{
  if (nsend == 0) return;
  const size_t n = (size_t)nsend * NCOMP * 4;
  k_halo_pack_upd<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(U0, R, dt, vol, send_elem, nsend, slab);
}

void launch_weno(int ndof, const DevMesh& m0, double cweight, const double* Uin, double* Uout,
                 hipStream_t s)
{
  if (m0.ne == 0 || ndof == 1) return;
  const DevMesh m = lim_range(m0);
  if (m.pde == 1) {
    QDG_DISPATCH_NDOF(ndof, (tr::k_weno<N><<<dim3(nblk(m.ne, 256), m.ncomp), 256, 0, s>>>(m, cweight, Uin, Uout)));
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
  if (m.pde == 1) {
    QDG_DISPATCH_NDOF(ndof, (tr::k_rk<N><<<(unsigned)(((size_t)m.nie * m.ncomp * N + 255) / 256), 256, 0, s>>>(m, a, b, dt, Un, R, U, Uout)));
    return;
  }
  QDG_DISPATCH_NDOF(ndof, (k_rk<N><<<(unsigned)(((size_t)m.nie * NCOMP * N + 255) / 256), 256, 0, s>>>(m, a, b, dt, Un, R, U, Uout)));
}

void launch_mass(int ndof, const DevMesh& m, double* L, hipStream_t s)
{
  if (m.ne == 0) return;
  if (m.pde == 1) {
    QDG_DISPATCH_NDOF(ndof, (tr::k_mass<N><<<nblk(m.ne, 256), 256, 0, s>>>(m, L)));
    return;
  }
  QDG_DISPATCH_NDOF(ndof, (k_mass<N><<<nblk(m.ne, 256), 256, 0, s>>>(m, L)));
}

void launch_init(int ndof, const DevMesh& m, const Phys& ph, double t, double* U, hipStream_t s)
{
  if (m.nie == 0) return;
  if (m.pde == 1) {
    QDG_DISPATCH_NDOF(ndof, (tr::k_init<N><<<dim3(nblk(m.nie, 256), m.ncomp), 256, 0, s>>>(m, ph, t, U)));
    return;
  }
  QDG_DISPATCH_NDOF(ndof, QDG_DISPATCH_PROB(ph.problem, (k_init<N, P><<<nblk(m.nie, 256), 256, 0, s>>>(m, ph, t, U))));
}

void launch_diag(int ndof, const DevMesh& m, const Phys& ph, double t_new, const double* U,
                 double* part, double* out, hipStream_t s)
{
  const int nb = nblk(m.nie, 256);
  if (nb > 0 && m.pde == 1) {
    // one grid row per scalar; part holds nb * ncomp block results (qdg_api.cpp sizes it)
    QDG_DISPATCH_NDOF(ndof, (tr::k_diag<N><<<dim3(nb, m.ncomp), 256, 0, s>>>(m, ph, t_new, U, part)));
    k_diag_final<<<1, 64, 0, s>>>(part, nb * m.ncomp, out);
    return;
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

void launch_rows_gather(size_t n, int nprop, const int* drow, const double* U, double* packed, hipStream_t s)
{
  if (n) k_rows_gather<<<(unsigned)((n * nprop + 255) / 256), 256, 0, s>>>(n, nprop, drow, U, packed);
}
void launch_rows_scatter(size_t n, int nprop, const int* drow, const double* packed, double* U, hipStream_t s)
{
  if (n) k_rows_scatter<<<(unsigned)((n * nprop + 255) / 256), 256, 0, s>>>(n, nprop, drow, packed, U);
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

__global__ __launch_bounds__(256) void k_tr_solution(Phys ph, int ncomp, int n, const double* __restrict__ x,
                                                     const double* __restrict__ y,
                                                     const double* __restrict__ z, double t,
                                                     double* __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int c = 0; c < ncomp; ++c) out[(size_t)i * ncomp + c] = tr::solution(ph, c, ncomp, x[i], y[i], z[i], t);
}

void launch_solution(int pde, int ncomp, const Phys& ph, int n, const double* x, const double* y, const double* z,
                     double t, double* out, hipStream_t s)
{
  if (n == 0) return;
  if (pde == 1) { k_tr_solution<<<nblk(n, 256), 256, 0, s>>>(ph, ncomp, n, x, y, z, t, out); return; }
  QDG_DISPATCH_PROB(ph.problem, (k_solution<P><<<nblk(n, 256), 256, 0, s>>>(ph, n, x, y, z, t, out)));
}

// dg::Transport::fieldOutput, src/PDE/Transport/DGTransport.hpp:248-279: per scalar c the mean,
// then per scalar Problem::solution at the centroid, then per scalar (analytic - numerical)^2 * vol
// (three blocks of ncomp fields, the order of fieldNames, DGTransport.hpp:211-246)
__global__ __launch_bounds__(256) void k_tr_field_output(DevMesh m, Phys ph, int ndof, double t,
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
  const int nc = m.ncomp;
  for (int c = 0; c < nc; ++c) {
    const double u = U[((size_t)e * nc + c) * ndof], sa = tr::solution(ph, c, nc, x, y, z, t);
    out[(size_t)c * nrows + h] = u;
    out[(size_t)(nc + c) * nrows + h] = sa;
    out[(size_t)(2 * nc + c) * nrows + h] = (sa - u) * (sa - u) * vol;
  }
}

int field_count(int pde, int ncomp, int problem)
{
  if (pde == 1) return 3 * ncomp;
  return problem == 3 ? 12 : problem == 4 ? 15 : problem == 7 ? 14 : problem == 10 ? 18 : problem == 0 ? 7 : 6;
}

// geoElem == nullptr: resident state in device rows, output permuted to the caller's
// numbering; else U and geoElem are in the caller's numbering (nrows rows)
void launch_field_output(int ndof, const DevMesh& m, const Phys& ph, double t, const double* U,
                         const double* geoElem, int nrows, double* out, hipStream_t s)
{
  if (nrows == 0) return;
  if (m.pde == 1) {
    k_tr_field_output<<<nblk(nrows, 256), 256, 0, s>>>(m, ph, ndof, t, U, geoElem, nrows, out);
  } else {
    QDG_DISPATCH_PROB(ph.problem, (k_field_output<P><<<nblk(nrows, 256), 256, 0, s>>>(m, ph, ndof, t, U, geoElem, nrows, out)));
  }
  if (!geoElem && m.ndofel)
    k_field_ndof<<<nblk(nrows, 256), 256, 0, s>>>(m, nrows, out + (size_t)field_count(m.pde, m.ncomp, ph.problem) * nrows);
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
