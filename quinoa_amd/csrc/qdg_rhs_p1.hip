// qdg_rhs_p1.hip -- the DG-P1 right-hand side of dg::CompFlow::rhs
// (src/PDE/CompFlow/DGCompFlow.hpp:130-195) for gfx950, hand-written HIP:
//   k_rhs_p1w  tile / face-task kernel, uniform order (the default and the headline kernel)
//   k_rhs_p1t  tile / face-task kernel of p-adaptive meshes (per-element ndof in {1, 4})
//   k_rhs_p1   element-centric form: every tet visits its four faces, R written once, bitwise
//              reproducible run to run (context option "p1_rhs" = 1)
#include "qdg_devfn.hpp"

namespace qdg {

// ------------------------------------------------- DG-P1 RHS, element-centric form
// One lane per tet, specialised for throughput:
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
__global__ __launch_bounds__(256, 2) void k_rhs_p1(DevMesh m, Phys ph, double t,
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


// PDG (p-adaptive DG, scheme pdg): a tet with m.ndofel == 1 is a P0 element --
// its state is its mean (Surface.cpp:146-156), only its mean is updated
// (update_rhs_fa, Surface.cpp:234-271), it has no volume term (Volume.cpp:56)
// and its source integral uses the 1-point rule (Source.cpp:54).  The face
// quadrature keeps 3 points where the reference takes max(ng_l, ng_r)
// (Surface.cpp:81-86): between two P0 tets both states are constant, so the 1-
// and the 3-point sums agree to rounding.
template <bool WITH_DT, bool FUSE_RK, int PROB, bool PDG>
__global__ __launch_bounds__(TILE_BS, 2) void k_rhs_p1t(DevMesh m, Phys ph, double t,
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
#pragma unroll 1
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
  // rows out through LDS as coalesced wave stores (see k_rhs_p1w)
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

// ------------------------------------------- the lean face task of the tile kernels below
// One face of a tile, evaluated once.  The two tets' vertex states at the face (2 x 15 LDS values)
// are read ONCE, at the top, and turned into the six point states at once (Y own side, X
// neighbour); nothing else is held across the Gauss points.  (Round 3 measured the forms: vertex
// states re-read per point 1.58 ms at 10.1 M tets, the neighbour's read once -- the reads that meet
// bank conflicts, partner tets are scattered over the tile -- 1.52 ms, both once 1.50 ms.)
// `a` is the packed task word, g4 the face record {area, n}, *nbrow the neighbour's device row
// (faces to other tiles; read only by lanes that have such a face -- they are listed first, so
// later rounds issue no load).  Everything is evaluated in the OWN tet's frame: left' =
// own, right' = neighbour, n' = the own tet's outward normal (the stored normal or its negative);
// for a face whose stored left tet is the neighbour this is the mirror image of the reference's
// evaluation (Sl' = -Sr, Sm' = -Sm, Sr' = -Sl), so HLLC's ladder (HLLC.hpp:93-124) is applied in
// its mirrored form (flux_hllc_own) -- same four fluxes, same fall-through of a NaN wave speed to
// the STORED right state -- and the own tet always loses what the neighbour gains.
// LDS layouts of a tile's vertex states / accumulators: idx(tet, vertex) is the word of component 0,
// components are CS words apart
struct LyTile {            // planes [vertex][component][tet] over the whole tile (k_rhs_p1w)
  static constexpr int CS = TILE;
  __device__ static __forceinline__ int idx(int e, int v) { return v * NCOMP * TILE + e; }
};

template <bool WITH_DT, int PROB>
__device__ __forceinline__ void face_task_lean(const DevMesh& m, const Phys& ph, double t,
                                               const double* __restrict__ U, double* __restrict__ nod,
                                               double* __restrict__ accN, double* __restrict__ sdelt,
                                               int a, const int* __restrict__ nbrow, int tile_e0,
                                               const double (&g4)[4])
{
  using LY = LyTile;
  constexpr int NDOF = 4, NGF = 3, NPROP = NCOMP * NDOF;
  constexpr bool HAS_DIRICHLET = (PROB == 3 || PROB == 4 || PROB == 0 || PROB == 7 || PROB == 10);
  const Tables<4>& T = c_tab4;
  const int el = TASK_EL(a), lf = TASK_LF(a), code = TASK_CODE(a), kind = TASK_KIND(a),
            bc = TASK_BC(a), pl = TASK_PL(a);
  const bool own_left = TASK_OWNLEFT(a);
  const double area = g4[0];
  const double osg = own_left ? 1.0 : -1.0;
  const double fn[3] = { osg * g4[1], osg * g4[2], osg * g4[3] };
  const bool bnd = kind == TASK_BND;
  // LDS word index of (tet, face vertex j, component 0); components are TILE words apart
  int ao[3], an[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    ao[j] = LY::idx(el, lpofa(lf, j));
    an[j] = LY::idx(pl, (code >> (2 * j)) & 3);
  }
  // X[g]: the neighbour's state at point g -- in-tile face: from its three vertex states in LDS,
  // read ONCE (the partner tets of a wave's lanes are scattered over the tile, so these are the
  // reads that meet bank conflicts; the own side's, consecutive tets, are conflict-free and are
  // read again per point); face to another tile: from the neighbour's modal row
  double Y[3][NCOMP], X[3][NCOMP];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) {
    const double v0 = nod[ao[0] + c * LY::CS], v1 = nod[ao[1] + c * LY::CS], v2 = nod[ao[2] + c * LY::CS];
    const double bo = (v0 + v1 + v2) * (1.0 / 6.0);
    Y[0][c] = fma(0.5, v1, bo); Y[1][c] = fma(0.5, v2, bo); Y[2][c] = fma(0.5, v0, bo);
  }
  if (kind == TASK_INT) {
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      const double v0 = nod[an[0] + c * LY::CS], v1 = nod[an[1] + c * LY::CS], v2 = nod[an[2] + c * LY::CS];
      const double bn = (v0 + v1 + v2) * (1.0 / 6.0);
      X[0][c] = fma(0.5, v1, bn); X[1][c] = fma(0.5, v2, bn); X[2][c] = fma(0.5, v0, bn);
    }
  } else if (kind == TASK_EXT) {
    double b1[3], b2[3], b3[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) vertex_basis((code >> (2 * j)) & 3, b1[j], b2[j], b3[j]);
    double rl[NCOMP][NDOF];
    load_row<NPROP>(U, *nbrow, &rl[0][0]);
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      const double x0 = rl[c][0], x1 = rl[c][1], x2 = rl[c][2], x3 = rl[c][3];
      const double v0 = x0 + x1 * b1[0] + x2 * b2[0] + x3 * b3[0];
      const double v1 = x0 + x1 * b1[1] + x2 * b2[1] + x3 * b3[1];
      const double v2 = x0 + x1 * b1[2] + x2 * b2[2] + x3 * b3[2];
      const double bn = (v0 + v1 + v2) * (1.0 / 6.0);
      X[0][c] = fma(0.5, v1, bn); X[1][c] = fma(0.5, v2, bn); X[2][c] = fma(0.5, v0, bn);
    }
  }
  const double wsel = (bnd && bc == 0) ? 0.0 : 1.0;   // boundary face without a BC: no flux
  const double refl = (bc == 2) ? 2.0 : 0.0;          // Symmetry: mirrored momentum
  [[maybe_unused]] ElemGeom gdir;
  if constexpr (HAS_DIRICHLET) {
    if (bnd && bc == 1) load_geom(m, tile_e0 + el, gdir);
  }

  // 3-point rule: state at point g = B + V_h(g)/2, h(g) = (g+1)%3; vertex-weighted flux sums
  // W_j = A/18 (F_0+F_1+F_2) + A/6 F_g(j), g(j) = (j+2)%3   (Quadrature.cpp:261-339)
  double Fg[3][NCOMP], dsum = 0.0;
#pragma unroll
  for (int ig = 0; ig < NGF; ++ig) {
    double so[NCOMP], sn[NCOMP];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) so[c] = Y[ig][c];
    if (kind != TASK_BND) {
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) sn[c] = X[ig][c];
    } else {
      // Extrapolate: u_r = u_l; Symmetry: mirrored momentum (DGCompFlow.hpp:672-690)
      const double vn2 = refl * (so[1] * fn[0] + so[2] * fn[1] + so[3] * fn[2]);
      sn[0] = so[0]; sn[1] = so[1] - vn2 * fn[0]; sn[2] = so[2] - vn2 * fn[1];
      sn[3] = so[3] - vn2 * fn[2]; sn[4] = so[4];
      if constexpr (HAS_DIRICHLET) {
        if (bc == 1) {
          double P[3];
          face_point(gdir, lf, T.fs[ig][0], T.fs[ig][1], T.fs[ig][2], P);
          prob_solution<PROB>(ph, P[0], P[1], P[2], t, sn);
        }
      }
    }
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
    if (ph.flux == 1) flux_lf_q(fn, so, sn, qo, qn, Fg[ig]);   // symmetric under the mirror image
    else flux_hllc_own(fn, so, sn, qo, qn, own_left, Fg[ig]);
    __builtin_amdgcn_sched_barrier(0);     // keep the three points in sequence (register pressure)
  }

  // scatter the vertex-weighted flux sums: the own tet loses, the neighbour gains
  const double k1 = area * wsel * (1.0 / 18.0), k2 = area * wsel * (1.0 / 6.0);
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) {
    const double Ssum = k1 * ((Fg[0][c] + Fg[1][c]) + Fg[2][c]);
    const double w0 = fma(k2, Fg[2][c], Ssum), w1 = fma(k2, Fg[0][c], Ssum), w2 = fma(k2, Fg[1][c], Ssum);
    __hip_atomic_fetch_add(accN + ao[0] + c * LY::CS, -w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(accN + ao[1] + c * LY::CS, -w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(accN + ao[2] + c * LY::CS, -w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (kind == TASK_INT) {
      __hip_atomic_fetch_add(accN + an[0] + c * LY::CS, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(accN + an[1] + c * LY::CS, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(accN + an[2] + c * LY::CS, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  if (WITH_DT) {
    dsum *= area * (1.0 / 3.0);
    __hip_atomic_fetch_add(sdelt + el, dsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (kind == TASK_INT)
      __hip_atomic_fetch_add(sdelt + pl, dsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

// phase 2 of the lean tile kernels, one lane per tet: face sums (modal form, tet_face_sums) + volume
// term (+ source), the volume states evaluated from the tile's vertex states in LDS -- the 5-point
// rule (Quadrature.cpp:16-259: the centroid with weight -4/5 and, per vertex v, the point with
// weight 1/2 on v and 1/6 on the others, 9/20 each) has the one-heavy-vertex structure of the
// face rule: state = SV/4 resp. SV/6 + V_v/3 with SV the sum of the four vertex states
// R[c][k] = sum_v accN[v][c] * B_k(vertex v): the face sums of a tet in modal form
__device__ __forceinline__ void tet_face_sums(const double* __restrict__ accN, int tl, double (&acc)[NCOMP][4])
{
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) {
    const double n0 = accN[LIDX(tl, 0, c)], n1 = accN[LIDX(tl, 1, c)], n2 = accN[LIDX(tl, 2, c)],
                 n3 = accN[LIDX(tl, 3, c)];
    acc[c][0] = (n0 + n1) + (n2 + n3);
    acc[c][1] = n1 - n0;
    acc[c][2] = 2.0 * n2 - n0 - n1;
    acc[c][3] = 3.0 * n3 - n0 - n1 - n2;
  }
}

// acc += volume term (+ source)
template <int PROB>
__device__ __forceinline__ void tet_volume_lean(const Phys& ph, double t, const double* __restrict__ nod,
                                                int tl, double vol, const ElemGeom& g, double (&acc)[NCOMP][4])
{
  constexpr int NDOF = 4;
  const Tables<4>& T = c_tab4;
  // the inverse Jacobian first: the twelve coordinates are dead before the flux sums are live
  double ji[3][3];
  inverse_jacobian(g, ji);
  // (an opaque use: keeps the compiler from sinking the computation to the contraction below,
  // which would carry the coordinates through the flux loop)
  asm volatile("" : "+v"(ji[0][0]), "+v"(ji[0][1]), "+v"(ji[0][2]), "+v"(ji[1][0]), "+v"(ji[1][1]),
                    "+v"(ji[1][2]), "+v"(ji[2][0]), "+v"(ji[2][1]), "+v"(ji[2][2]));
  double SV[NCOMP], Fs[NCOMP][3];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) {
    SV[c] = (nod[LIDX(tl, 0, c)] + nod[LIDX(tl, 1, c)]) + (nod[LIDX(tl, 2, c)] + nod[LIDX(tl, 3, c)]);
    Fs[c][0] = Fs[c][1] = Fs[c][2] = 0.0;
  }
#pragma unroll
  for (int ig = 0; ig < 5; ++ig) {
    double s[NCOMP];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c)
      s[c] = (ig == 0) ? 0.25 * SV[c] : fma(1.0 / 3.0, nod[LIDX(tl, (ig + 3) & 3, c)], SV[c] * (1.0 / 6.0));
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
  // acc[c][k] += vol * F_c . grad B_k
#pragma unroll
  for (int k = 1; k < NDOF; ++k) {
    const double g0 = T.vdB[0][0][k], g1 = T.vdB[0][1][k], g2 = T.vdB[0][2][k];
    const double dx = vol * (g0 * ji[0][0] + g1 * ji[1][0] + g2 * ji[2][0]);
    const double dy = vol * (g0 * ji[0][1] + g1 * ji[1][1] + g2 * ji[2][1]);
    const double dz = vol * (g0 * ji[0][2] + g1 * ji[1][2] + g2 * ji[2][2]);
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) acc[c][k] += Fs[c][0] * dx + Fs[c][1] * dy + Fs[c][2] * dz;
  }
  if constexpr (prob_has_source<PROB>()) {
#pragma unroll 1
    for (int ig = 0; ig < 5; ++ig) {
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
}

// SSP-RK3 update fused into the epilogue: acc <- a*Un + b*(U + dt*acc/L), rows in registers
__device__ __forceinline__ void rk_epilogue_rows(const double (&u)[NCOMP][4], const double (&un)[NCOMP][4],
                                                 double dtv, double rk_a, double rk_b, double (&acc)[NCOMP][4])
{
  constexpr double imf[4] = { 1.0, 10.0, 10.0 / 3.0, 5.0 / 3.0 };
#pragma unroll
  for (int c = 0; c < NCOMP; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[c][k] = rk_a * un[c][k] + rk_b * (u[c][k] + dtv * imf[k] * acc[c][k]);
}

// ------------------------------------------- DG-P1 RHS, tile / face-task form (uniform order)
// A workgroup of 256 lanes owns a tile of TILE = 248 consecutive device rows (Morton-compact),
// two workgroups per CU (2 x 79 KB of LDS), in three phases separated by two barriers:
//  0. every load that depends on nothing goes out at kernel entry in one burst (the tet's own
//     modal row, the task words and partner-row ids of all rounds, round 0's face record, node ids, volume); the row
//     is written to LDS in NODAL form, nod[vertex][c][tet], and STAYS in registers for the RK
//     epilogue (re-read in phase 2 it misses the L2 at size: 160 B per tet of extra HBM reads);
//  1. one lane per face task (face_task_lean): every face of the tile once, both tets of an
//     in-tile face served by one evaluation through ds_add_f64;
//  2. one lane per tet (tet_volume_lean): accumulators -> modal R, volume term from the vertex
//     states in LDS, source; WITH_DT: CFL sum -> vol/sum, block minimum (stage 0); FUSE_RK:
//     Uout = a*Un + b*(U + dt*R/L) written instead of R (stages 1, 2); rows leave through LDS as
//     coalesced, non-temporal 1-KiB wave stores.
// Round 3 measured the alternatives on this kernel (profiles/r03_p1_experiments.log): 384 / 320
// lanes per tile get ONE workgroup per CU from the dispatcher (2.7 ms against 1.6 ms at 10.1 M
// tets), 512 lanes need 128 registers (spills: scratch traffic costs more than the waves hide),
// 160-row tiles with three workgroups per CU at 158 registers 1.67 ms, persistent workgroups
// prefetching the next tile's rows into registers through the face rounds 2.1 ms (256 registers,
// in-order vmcnt couples the prefetch to every later load) or by LDS-DMA into the accumulator
// planes during phase 2 1.84 ms, the round-2 form of the face task (vertex states held across
// the Gauss points, 206 registers, the row re-read in phase 2) 1.60-1.65 ms.  Round 4 repeated the 160-row
// experiment with a register-lean fused epilogue (three workgroups per CU for both instantiations): a wash
// (profiles/r04_p1_experiments.log).
template <bool WITH_DT, bool FUSE_RK, int PROB>
__global__ __launch_bounds__(TILE_BS, 2) void k_rhs_p1w(DevMesh m, Phys ph, double t,
                                                        const double* __restrict__ U,
                                                        double* __restrict__ R,
                                                        double* __restrict__ blockmin,
                                                        double rk_a, double rk_b,
                                                        const double* __restrict__ dtp,
                                                        const double* __restrict__ Un)
{
  constexpr int NDOF = 4, NPROP = NCOMP * NDOF, BS = TILE_BS;
  constexpr int NR = 4;                               // rounds of the workgroup over the padded task slots
  static_assert(TILE <= TILE_BS, "task_stride = 4 * TILE_BS slots per tile: 4 rounds of the workgroup's lanes");
  __shared__ __attribute__((aligned(16))) double nod[TILE * NPROP];
  __shared__ double accN[TILE * NPROP];
  __shared__ double sdelt[WITH_DT ? TILE : 1];
  const int tid = threadIdx.x;
  const int tile = m.blk0 + xcd_tile(blockIdx.x, gridDim.x);
  const int tile_e0 = tile * TILE;
  const int nloc = (m.nie - tile_e0 < TILE) ? m.nie - tile_e0 : TILE;

  // kernel entry: everything that depends on nothing
  const size_t slot0 = (size_t)tile * (4 * TILE_BS) + tid;
  double r[NCOMP][NDOF];
  const int erow = tile_e0 + ((tid < nloc) ? tid : 0);          // lanes beyond the tile read its row 0
  load_row<NPROP>(U, erow, &r[0][0]);
  int ta[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) ta[q] = m.task_a[slot0 + BS * q];
  // the partner rows' ids of the four rounds' tasks too: read inside a round, the id is one more dependent round
  // trip in front of the partner row's (RHS launch 1508 / 1500 -> 1499 / 1471 us at 10.1 M tets)
  int tb[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) tb[q] = m.task_nb[slot0 + BS * q];
  double gnx[4];
  load_row<4>(m.tgeo, slot0, gnx);                               // (zeros behind unused slots)
  int in4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) in4[i] = m.inpoel[(size_t)i * m.stride + erow];
  const double vol = m.vol[erow];

  // ---- phase 0: modal row -> the 4 vertex states; accumulators = 0 ---------------------
  if (tid < TILE) {
    if (tid >= nloc) {
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) { r[c][0] = 1.0; r[c][1] = r[c][2] = r[c][3] = 0.0; }
    }
    // LDS planes [vertex][component][tet]: lanes of a wave work on different tets at the same
    // (vertex, component), so tet-fastest storage is free of bank conflicts
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      // B at the vertices: v0 (-1,-1,-1), v1 (1,-1,-1), v2 (0,2,-1), v3 (0,0,3)
      const double a = r[c][0] - r[c][3];
      nod[LIDX(tid, 0, c)] = a - r[c][1] - r[c][2];
      nod[LIDX(tid, 1, c)] = a + r[c][1] - r[c][2];
      nod[LIDX(tid, 2, c)] = a + 2.0 * r[c][2];
      nod[LIDX(tid, 3, c)] = r[c][0] + 3.0 * r[c][3];
#pragma unroll
      for (int vx = 0; vx < 4; ++vx) accN[LIDX(tid, vx, c)] = 0.0;
    }
    if (WITH_DT) sdelt[tid] = 0.0;
  }
  __syncthreads();

  // ---- phase 1: one lane per face task ------------------------------------------
#pragma unroll 1
  for (int q = 0; q < NR; ++q) {
    const int a = (q == 0) ? ta[0] : (q == 1) ? ta[1] : (q == 2) ? ta[2] : ta[3];
    if (a < 0) break;
    const double g4[4] = { gnx[0], gnx[1], gnx[2], gnx[3] };
    const int an_ = (q == 0) ? ta[1] : (q == 1) ? ta[2] : (q == 2) ? ta[3] : -1;
    if (an_ >= 0) load_row<4>(m.tgeo, slot0 + (size_t)BS * (q + 1), gnx);    // the next round's face record
    const int nbr_row = (q == 0) ? tb[0] : (q == 1) ? tb[1] : (q == 2) ? tb[2] : tb[3];
    face_task_lean<WITH_DT, PROB>(m, ph, t, U, nod, accN, sdelt, a, &nbr_row, tile_e0, g4);
  }

  // phase-2 inputs are requested before the barrier (the node ids are here already)
  ElemGeom g;
  [[maybe_unused]] double un[NCOMP][NDOF];
  if (tid < nloc) {
    if constexpr (FUSE_RK) load_row<NPROP>(Un, tile_e0 + tid, &un[0][0]);
    double q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      load_row<4>(m.xyz4, in4[i], q);
      g.p[i][0] = q[0]; g.p[i][1] = q[1]; g.p[i][2] = q[2];
    }
  }
  __syncthreads();

  // ---- phase 2: one lane per tet: volume (+source) term, epilogue, store --------
  double dte = DBL_MAX;
  double acc[NCOMP][NDOF];
  if (tid < nloc) {
    tet_face_sums(accN, tid, acc);
    tet_volume_lean<PROB>(ph, t, nod, tid, vol, g, acc);
    if constexpr (FUSE_RK) {
      rk_epilogue_rows(r, un, dtp[0] / vol, rk_a, rk_b, acc);
      halo_fold_row<NPROP>(m, tile_e0 + tid, &acc[0][0]);      // (qdg_step_comm: the next comsol pack, folded in)
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
      const int i = j * BS + tid;
      // non-temporal: the rows are read next by another kernel, after gigabytes of other traffic, and must not
      // displace the partner rows that the cross-tile face tasks of this launch find in the L2 (RHS launch
      // 1.54 -> 1.51 ms at 10.1 M tets; at 1 M the update + limiter kernel that follows gains 6 %)
      if (i < nvalid) store_nt(dst + i, src[i]);
    }
  }

  if (WITH_DT) {
    for (int off = 32; off > 0; off >>= 1) dte = fmin(dte, __shfl_down(dte, off, 64));
    __shared__ double wmin[(BS + 63) / 64];
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) wmin[wv] = dte;
    __syncthreads();
    if (tid == 0) {
      double mn = wmin[0];
      for (int w = 1; w < (BS + 63) / 64; ++w) mn = fmin(mn, wmin[w]);
      blockmin[tile] = mn;
    }
  }
}

// ================================================================ launchers


hipError_t upload_tables_p1(const Tables<1>& t1, const Tables<4>& t4, const Tables<10>& t10,
                            const QuadTet* qinit, const QuadTet* qdiag)
{
  return upload_tables_here(t1, t4, t10, qinit, qdiag);
}

// element-centric form (bitwise reproducible; option p1_rhs = 1); with_dt: also reduce
// min(vol/delt) into out_raw/out_dt
void launch_rhs_p1(const DevMesh& m, const Phys& ph, double t, const double* U, double* R,
                   bool with_dt, double* blockmin, double scale, double tleft, double* out_raw,
                   double* out_dt, hipStream_t s)
{
  const int nb = nblk(m.nie, 256);
  if (nb == 0) return;
  if (with_dt) {
    QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1<true, false, P><<<nb, 256, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr)));
    launch_dt_final(blockmin, nb, scale, tleft, out_raw, out_dt, s);
  } else {
    QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1<false, false, P><<<nb, 256, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr)));
  }
}

// tile / face-task form of the P1 RHS; tiles [first, first+count) (count < 0: all).
// Uniform order runs k_rhs_p1w, p-adaptive meshes (m.ndofel) k_rhs_p1t.  With with_dt the
// launch that ends at the last tile also reduces the per-tile minima to the time step.
void launch_rhs_p1t(const DevMesh& m0, const Phys& ph, double t, const double* U, double* R,
                    bool with_dt, double* blockmin, double scale, double tleft, double* out_raw,
                    double* out_dt, hipStream_t s, int first, int count)
{
  if (m0.ntile == 0) return;
  DevMesh m = m0;
  m.blk0 = first;
  const int nb = count < 0 ? m.ntile - first : count;
  if (nb > 0 && !m.ndofel) {
    if (with_dt) {
      QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1w<true, false, P><<<nb, TILE_BS, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr)));
    } else {
      QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1w<false, false, P><<<nb, TILE_BS, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr)));
    }
  } else if (nb > 0) {
    if (with_dt) {
      QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1t<true, false, P, true><<<nb, TILE_BS, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr)));
    } else {
      QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1t<false, false, P, true><<<nb, TILE_BS, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr)));
    }
  }
  if (with_dt && first + nb == m.ntile)
    launch_dt_final(blockmin, m.ntile, scale, tleft, out_raw, out_dt, s);
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
  if (!m.ndofel) {
    QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1w<false, true, P><<<nb, TILE_BS, 0, s>>>(m, ph, t, U, Uout, nullptr, a, b, dt, Un)));
    return;
  }
  QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1t<false, true, P, true><<<nb, TILE_BS, 0, s>>>(m, ph, t, U, Uout, nullptr, a, b, dt, Un)));
}

// element-centric form with the SSP-RK3 update fused in: Uout = a*Un + b*(U + dt*R/L)
void launch_rhs_p1_rk(const DevMesh& m, const Phys& ph, double t, const double* U, double* Uout,
                      double a, double b, const double* dt, const double* Un, hipStream_t s)
{
  const int nb = nblk(m.nie, 256);
  if (nb == 0) return;
  QDG_DISPATCH_PROB(ph.problem, (k_rhs_p1<false, true, P><<<nb, 256, 0, s>>>(m, ph, t, U, Uout, nullptr, a, b, dt, Un)));
}

}  // namespace qdg
