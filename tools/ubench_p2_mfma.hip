// ubench_p2_mfma.hip -- does the f64 matrix core pay for the DG-P2 basis contraction?
//
// BASELINE config 3 names "MFMA P2 contraction".  The one GEMM-shaped piece of the P2 RHS with
// a SHARED operand is the evaluation of a tet's own state at its 24 face + 11 volume Gauss
// points from its 10 modes (tk::eval_state over the constant basis table, Basis.cpp:290-302):
//     S[t][c][g] = sum_k U[t][c][k] * B[k][g],   t tets, c = 5, k = 10, g = 35.
// This program times exactly that contraction, 1 M tets, two ways:
//   valu : one lane per tet, row in registers, B through scalar loads (what k_rhs<10> does)
//   mfma : rows staged in LDS, v_mfma_f64_16x16x4_f64 (16 tets x 16 points x 4 modes per
//          instruction; K padded 10 -> 12, N padded 35 -> 48), A operands read from LDS in the
//          instruction's layout, D consumed in place
// Both feed every S into the same cheap consumer (sum_g w_g S^2 per tet and component) so that
// nothing is optimised away and nothing but 40 B per tet is written.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_p2_mfma.hip -o tools/ubench_p2_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NK = 10, NG = 35, NC = 5, NPROP = NC * NK;
__constant__ double c_B[NK][NG];
__constant__ double c_w[NG];

// ---------------------------------------------------------------- VALU: lane per tet
__global__ __launch_bounds__(256) void k_valu(int ntet, const double* __restrict__ U, double* __restrict__ out)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntet) return;
  double u[NC][NK];
  const double2* p = reinterpret_cast<const double2*>(U + (size_t)t * NPROP);
#pragma unroll
  for (int i = 0; i < NPROP / 2; ++i) { const double2 v = p[i]; (&u[0][0])[2 * i] = v.x; (&u[0][0])[2 * i + 1] = v.y; }
  double acc[NC] = { 0, 0, 0, 0, 0 };
#pragma unroll 1
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < NK; ++k) s = fma(u[c][k], c_B[k][g], s);
      acc[c] = fma(c_w[g] * s, s, acc[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) out[(size_t)t * NC + c] = acc[c];
}

// ---------------------------------------------------------------- MFMA: 256 tets per workgroup
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mfma(int ntet, const double* __restrict__ U, double* __restrict__ out)
{
  __shared__ double rows[256 * NPROP];              // 100 KiB: the tile's rows, row-major
  __shared__ double red[256 * NC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t0 = blockIdx.x * 256;
  {
    const double2* src = reinterpret_cast<const double2*>(U + (size_t)t0 * NPROP);
    double2* dst = reinterpret_cast<double2*>(rows);
    const int nvalid = ((ntet - t0 < 256) ? ntet - t0 : 256) * (NPROP / 2);
#pragma unroll
    for (int j = 0; j < NPROP / 2; ++j) {
      const int i = j * 256 + tid;
      dst[i] = (i < nvalid) ? src[i] : make_double2(0.0, 0.0);
    }
  }
  // B operand of the instruction: lane holds B[k = 4*kb + (lane>>4)][g = 16*nb + (lane&15)]
  double bop[3][3], wcol[3];
#pragma unroll
  for (int nb = 0; nb < 3; ++nb) {
    const int g = 16 * nb + (lane & 15);
    wcol[nb] = (g < NG) ? c_w[g] : 0.0;
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
      const int k = 4 * kb + (lane >> 4);
      bop[kb][nb] = (k < NK && g < NG) ? c_B[k][g] : 0.0;
    }
  }
  __syncthreads();
  // each wave: its 64 tets in 4 blocks of 16; per block and component 9 MFMAs
#pragma unroll 1
  for (int blk = 0; blk < 4; ++blk) {
    const int trow = wave * 64 + blk * 16 + (lane & 15);        // A: row i = lane & 15
#pragma unroll 1
    for (int c = 0; c < NC; ++c) {
      double aop[3];
#pragma unroll
      for (int kb = 0; kb < 3; ++kb) {
        const int k = 4 * kb + (lane >> 4);                      // A: k index = lane >> 4
        aop[kb] = rows[trow * NPROP + c * NK + (k < NK ? k : NK - 1)];   // padded k: B holds 0
      }
      double part[4] = { 0, 0, 0, 0 };                           // rows (lane>>4) + 4 r of the block
#pragma unroll
      for (int nb = 0; nb < 3; ++nb) {
        v4d d = { 0, 0, 0, 0 };
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) d = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[kb], bop[kb][nb], d, 0, 0, 0);
        // D: col = lane & 15 (point), row = (lane >> 4) + 4 r (tet of the block)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[r] = fma(wcol[nb] * d[r], d[r], part[r]);
      }
      // sum over the 16 points = over the 16 lanes of a row group
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double v = part[r];
        for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 16);
        if ((lane & 15) == 0) red[(wave * 64 + blk * 16 + (lane >> 4) + 4 * r) * NC + c] = v;
      }
    }
  }
  __syncthreads();
  const int t = t0 + tid;
  if (t < ntet) {
#pragma unroll
    for (int c = 0; c < NC; ++c) out[(size_t)t * NC + c] = red[tid * NC + c];
  }
}

int main(int argc, char** argv)
{
  const int ntet = argc > 1 ? atoi(argv[1]) : 998250;
  std::vector<double> hB(NK * NG), hw(NG), hU((size_t)ntet * NPROP);
  for (int k = 0; k < NK; ++k) for (int g = 0; g < NG; ++g) hB[k * NG + g] = std::cos(0.37 * k + 0.11 * g) + (k == 0);
  for (int g = 0; g < NG; ++g) hw[g] = 1.0 / NG + 0.001 * g;
  srand(7);
  for (auto& v : hU) v = rand() / (double)RAND_MAX - 0.5;
  CHK(hipMemcpyToSymbol(HIP_SYMBOL(c_B), hB.data(), sizeof(double) * NK * NG));
  CHK(hipMemcpyToSymbol(HIP_SYMBOL(c_w), hw.data(), sizeof(double) * NG));
  double *dU, *o1, *o2;
  CHK(hipMalloc(&dU, hU.size() * 8)); CHK(hipMalloc(&o1, (size_t)ntet * NC * 8)); CHK(hipMalloc(&o2, (size_t)ntet * NC * 8));
  CHK(hipMemcpy(dU, hU.data(), hU.size() * 8, hipMemcpyHostToDevice));
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  const int nb = (ntet + 255) / 256, rep = 20;
  float ms[2];
  for (int which = 0; which < 2; ++which) {
    for (int i = 0; i < 3; ++i) { if (which) k_mfma<<<nb, 256>>>(ntet, dU, o2); else k_valu<<<nb, 256>>>(ntet, dU, o1); }
    CHK(hipEventRecord(a));
    for (int i = 0; i < rep; ++i) { if (which) k_mfma<<<nb, 256>>>(ntet, dU, o2); else k_valu<<<nb, 256>>>(ntet, dU, o1); }
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    CHK(hipEventElapsedTime(&ms[which], a, b));
    ms[which] /= rep;
  }
  std::vector<double> h1((size_t)ntet * NC), h2((size_t)ntet * NC);
  CHK(hipMemcpy(h1.data(), o1, h1.size() * 8, hipMemcpyDeviceToHost));
  CHK(hipMemcpy(h2.data(), o2, h2.size() * 8, hipMemcpyDeviceToHost));
  double err = 0.0, mx = 0.0;
  for (size_t i = 0; i < h1.size(); ++i) { err = std::fmax(err, std::fabs(h1[i] - h2[i])); mx = std::fmax(mx, std::fabs(h1[i])); }
  const double fl = 2.0 * ntet * NC * NK * NG;      // useful flops of the contraction
  printf("{\"ntet\": %d, \"useful_gflop\": %.3f, \"valu_ms\": %.4f, \"valu_tflops\": %.2f, \"mfma_ms\": %.4f, "
         "\"mfma_tflops_useful\": %.2f, \"mfma_over_valu_time\": %.3f, \"max_rel_diff\": %.2e, "
         "\"bytes_read_MB\": %.1f}\n",
         ntet, fl / 1e9, ms[0], fl / ms[0] / 1e9, ms[1], fl / ms[1] / 1e9, ms[1] / ms[0], err / mx,
         ntet * (double)NPROP * 8 / 1e6);
  return 0;
}
