// qdg_rhs_p2.hip -- the DG-P2 right-hand side of dg::CompFlow::rhs
// (src/PDE/CompFlow/DGCompFlow.hpp:130-195) for gfx950, hand-written HIP: a lane pair per tet.
#include "qdg_devfn.hpp"

namespace qdg {

__device__ P2Split g_p2s;          // copied to LDS by every workgroup of k_rhs_p2s

// ------------------------------------------------- DG-P2 RHS, two lanes per tet
// A one-lane-per-tet form needs the tet's row, its accumulators and a neighbour row
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

  // ---- source and volume integrals, two points per step: this lane's and its partner's ----
  // The source term (a function of position and time alone) has a loop of its own IN FRONT of the
  // flux loop: it needs the tet's twelve node coordinates, the flux loop their inverse Jacobian and
  // 60 registers of contracted fluxes -- together they spilled 88 bytes per lane to scratch, and a
  // kernel with scratch runs at a fraction of its speed on this chip (round 3).
  {
    ElemGeom g;
    load_geom(m, e, g);
    if constexpr (prob_has_source<PROB>()) {
      // src/PDE/Integrate/Source.cpp:21-141
#pragma unroll 1
      for (int s = 0; s < 6; ++s) {
        const int gm = 2 * s + h, gp = 2 * s + 1 - h;
        const double* tm = S.vol[gm][h];
        const double* tp = S.vol[gp][h];
        const double wt = S.vw[gm] * vol;
        const double xi = S.vc[gm][0], eta = S.vc[gm][1], zeta = S.vc[gm][2];
        const double w0 = 1.0 - xi - eta - zeta;
        double P[3], sr[NCOMP];
#pragma unroll
        for (int d = 0; d < 3; ++d)
          P[d] = g.p[0][d] * w0 + g.p[1][d] * xi + g.p[2][d] * eta + g.p[3][d] * zeta;
        prob_src<PROB>(ph, P[0], P[1], P[2], t, sr);
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
          if (prob_src_is_zero<PROB>(c)) continue;
          const double wm = wt * sr[c];
          const double wp = pair_swap(wm);
#pragma unroll
          for (int k = 0; k < KH; ++k) acc[c][k] += wm * tm[k] + wp * tp[k];
        }
      }
    }
    double ji[3][3];
    inverse_jacobian(g, ji);
    // (an opaque use: the coordinates are dead from here on)
    asm volatile("" : "+v"(ji[0][0]), "+v"(ji[0][1]), "+v"(ji[0][2]), "+v"(ji[1][0]), "+v"(ji[1][1]),
                      "+v"(ji[1][2]), "+v"(ji[2][0]), "+v"(ji[2][1]), "+v"(ji[2][2]));
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
      if (i < nvalid) store_nt(dst + i, src[i]);      // (non-temporal: qdg_devfn.hpp, store_nt)
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

// ================================================================ launchers

// this translation unit's constant tables plus the LDS-staged tables of k_rhs_p2s, built from
// the same rules and basis functions as Tables<10>
hipError_t upload_tables_p2(const Tables<1>& t1, const Tables<4>& t4, const Tables<10>& t10,
                            const QuadTet* qinit, const QuadTet* qdiag)
{
  hipError_t e = upload_tables_here(t1, t4, t10, qinit, qdiag);
  if (e != hipSuccess) return e;
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
  return hipMemcpyToSymbol(HIP_SYMBOL(g_p2s), &ps, sizeof(ps));
}

// mode 0: R = rhs(U); 1: + per-workgroup minima of vol/delt into blockmin (stage 0 with a CFL
// time step; the caller finishes with launch_dt_final over p2_rhs_blocks(m) values); 2: the
// SSP-RK3 update fused in, R <- a*Un + b*(U + dt*rhs/L)
int p2_rhs_blocks(const DevMesh& m) { return nblk(m.nie, 128); }

void launch_rhs_p2(const DevMesh& m, const Phys& ph, double t, const double* U, double* R, int mode,
                   double* blockmin, double a, double b, const double* dt, const double* Un, hipStream_t s)
{
  const int nb = p2_rhs_blocks(m);
  if (nb == 0) return;
  if (mode == 0) {
    QDG_DISPATCH_PROB(ph.problem, (k_rhs_p2s<P, 0><<<nb, 256, 0, s>>>(m, ph, t, U, R, nullptr, 0.0, 0.0, nullptr, nullptr)));
  } else if (mode == 1) {
    QDG_DISPATCH_PROB(ph.problem, (k_rhs_p2s<P, 1><<<nb, 256, 0, s>>>(m, ph, t, U, R, blockmin, 0.0, 0.0, nullptr, nullptr)));
  } else {
    QDG_DISPATCH_PROB(ph.problem, (k_rhs_p2s<P, 2><<<nb, 256, 0, s>>>(m, ph, t, U, R, nullptr, a, b, dt, Un)));
  }
}

}  // namespace qdg
