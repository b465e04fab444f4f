// ubench_rowstream.hip -- what does "one lane streams its own 160-byte row" cost against fully
// coalesced wave accesses staged through LDS?
//
// Every streaming kernel of the DG-P1 step (k_superbee, k_upd_superbee, k_rk, phase 0 / phase 2 of
// k_rhs_p1v) lets lane e read row e (20 doubles) as 10 x 16 B at a 160-B lane stride: each wave
// instruction touches 64 different 128-B lines.  This program measures, on rows far larger than
// the caches, the rate of
//   row     : that access as the kernels issue it today
//   lds     : the workgroup's 256 consecutive rows read as 10 fully coalesced 4-KiB wave... (16 B per
//             lane, 1 KiB contiguous per wave instruction), staged in LDS, each lane then reads its
//             row from LDS (row stride padded to 168 B: conflict-free 8-byte reads)
// for a read-only sweep (sum of the row -> 8 B per row), a write-only sweep and a copy a -> b with
// one FMA per element (the k_rk shape, two input streams and one output).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_rowstream.hip -o tools/ubench_rowstream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NP = 20;           // doubles per row
constexpr int BS = 256;
constexpr int LSTR = 21;         // LDS row stride in doubles (168 B)

// ---------------------------------------------------------------- lane-per-row
__global__ __launch_bounds__(BS) void k_read_row(int n, const double* __restrict__ U, double* __restrict__ out)
{
  const int e = blockIdx.x * BS + threadIdx.x;
  if (e >= n) return;
  const double2* p = reinterpret_cast<const double2*>(U + (size_t)e * NP);
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NP / 2; ++i) { const double2 v = p[i]; s += v.x + v.y; }
  out[e] = s;
}

__global__ __launch_bounds__(BS) void k_write_row(int n, double* __restrict__ U, const double* __restrict__ in)
{
  const int e = blockIdx.x * BS + threadIdx.x;
  if (e >= n) return;
  const double s = in[e];
  double2* p = reinterpret_cast<double2*>(U + (size_t)e * NP);
#pragma unroll
  for (int i = 0; i < NP / 2; ++i) p[i] = make_double2(s + i, s - i);
}

__global__ __launch_bounds__(BS) void k_axpy_row(int n, const double* __restrict__ A, const double* __restrict__ B,
                                                 double* __restrict__ C, double a, double b)
{
  const int e = blockIdx.x * BS + threadIdx.x;
  if (e >= n) return;
  const double2* pa = reinterpret_cast<const double2*>(A + (size_t)e * NP);
  const double2* pb = reinterpret_cast<const double2*>(B + (size_t)e * NP);
  double2* pc = reinterpret_cast<double2*>(C + (size_t)e * NP);
  double2 va[NP / 2], vb[NP / 2];
#pragma unroll
  for (int i = 0; i < NP / 2; ++i) { va[i] = pa[i]; vb[i] = pb[i]; }
#pragma unroll
  for (int i = 0; i < NP / 2; ++i) pc[i] = make_double2(a * va[i].x + b * vb[i].x, a * va[i].y + b * vb[i].y);
}

// ---------------------------------------------------------------- coalesced through LDS
// the workgroup's rows [e0, e0 + BS) are BS*NP/2 = 2560 double2; thread t moves double2 j*BS + t
__device__ __forceinline__ void rows_in(const double* __restrict__ U, int e0, int nrow, double* lds)
{
  const double2* src = reinterpret_cast<const double2*>(U + (size_t)e0 * NP);
  const int nval = nrow * (NP / 2);
#pragma unroll
  for (int j = 0; j < NP / 2; ++j) {
    const int i = j * BS + threadIdx.x;
    if (i < nval) {
      const double2 v = src[i];
      const int r = i / (NP / 2), p = i - r * (NP / 2);
      lds[r * LSTR + 2 * p] = v.x;
      lds[r * LSTR + 2 * p + 1] = v.y;
    }
  }
}
__device__ __forceinline__ void rows_out(double* __restrict__ U, int e0, int nrow, const double* lds)
{
  double2* dst = reinterpret_cast<double2*>(U + (size_t)e0 * NP);
  const int nval = nrow * (NP / 2);
#pragma unroll
  for (int j = 0; j < NP / 2; ++j) {
    const int i = j * BS + threadIdx.x;
    if (i < nval) {
      const int r = i / (NP / 2), p = i - r * (NP / 2);
      dst[i] = make_double2(lds[r * LSTR + 2 * p], lds[r * LSTR + 2 * p + 1]);
    }
  }
}

__global__ __launch_bounds__(BS) void k_read_lds(int n, const double* __restrict__ U, double* __restrict__ out)
{
  __shared__ double lds[BS * LSTR];
  const int e0 = blockIdx.x * BS, nrow = (n - e0 < BS) ? n - e0 : BS;
  rows_in(U, e0, nrow, lds);
  __syncthreads();
  if ((int)threadIdx.x < nrow) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) s += lds[threadIdx.x * LSTR + i];
    out[e0 + threadIdx.x] = s;
  }
}

__global__ __launch_bounds__(BS) void k_write_lds(int n, double* __restrict__ U, const double* __restrict__ in)
{
  __shared__ double lds[BS * LSTR];
  const int e0 = blockIdx.x * BS, nrow = (n - e0 < BS) ? n - e0 : BS;
  if ((int)threadIdx.x < nrow) {
    const double s = in[e0 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < NP / 2; ++i) { lds[threadIdx.x * LSTR + 2 * i] = s + i; lds[threadIdx.x * LSTR + 2 * i + 1] = s - i; }
  }
  __syncthreads();
  rows_out(U, e0, nrow, lds);
}

// element-wise: no transposition needed at all when the operation is per element (k_rk's case)
__global__ __launch_bounds__(BS) void k_axpy_flat(size_t n2, const double2* __restrict__ A, const double2* __restrict__ B,
                                                  double2* __restrict__ C, double a, double b)
{
  const size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
  if (i >= n2) return;
  const double2 va = A[i], vb = B[i];
  C[i] = make_double2(a * va.x + b * vb.x, a * va.y + b * vb.y);
}

// per-row work on both sides of the LDS: read rows coalesced, lane-per-row compute, write coalesced
__global__ __launch_bounds__(BS) void k_axpy_lds(int n, const double* __restrict__ A, const double* __restrict__ B,
                                                 double* __restrict__ C, double a, double b)
{
  __shared__ double la[BS * LSTR];
  __shared__ double lb[BS * LSTR];
  const int e0 = blockIdx.x * BS, nrow = (n - e0 < BS) ? n - e0 : BS;
  rows_in(A, e0, nrow, la);
  rows_in(B, e0, nrow, lb);
  __syncthreads();
  if ((int)threadIdx.x < nrow) {
#pragma unroll
    for (int i = 0; i < NP; ++i) la[threadIdx.x * LSTR + i] = a * la[threadIdx.x * LSTR + i] + b * lb[threadIdx.x * LSTR + i];
  }
  __syncthreads();
  rows_out(C, e0, nrow, la);
}

template <class F> static float timeit(F&& f, int rep)
{
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CHK(hipEventRecord(a));
  for (int i = 0; i < rep; ++i) f();
  CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
  float ms; CHK(hipEventElapsedTime(&ms, a, b));
  return ms / rep;
}

int main(int argc, char** argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 10110954;
  const size_t nd = (size_t)n * NP;
  double *A, *B, *C, *s;
  CHK(hipMalloc(&A, nd * 8)); CHK(hipMalloc(&B, nd * 8)); CHK(hipMalloc(&C, nd * 8)); CHK(hipMalloc(&s, (size_t)n * 8));
  {
    std::vector<double> h(nd);
    for (size_t i = 0; i < nd; ++i) h[i] = (double)(i % 1013) * 1e-3;
    CHK(hipMemcpy(A, h.data(), nd * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(B, h.data(), nd * 8, hipMemcpyHostToDevice));
    CHK(hipMemset(s, 0, (size_t)n * 8));
  }
  const int nb = (n + BS - 1) / BS, rep = 20;
  const double rowMB = nd * 8 / 1e6, sMB = n * 8.0 / 1e6;
  struct { const char* name; float ms; double MB; } r[8];
  int k = 0;
  r[k++] = { "read_row",   timeit([&] { k_read_row<<<nb, BS>>>(n, A, s); }, rep), rowMB + sMB };
  r[k++] = { "read_lds",   timeit([&] { k_read_lds<<<nb, BS>>>(n, A, s); }, rep), rowMB + sMB };
  r[k++] = { "write_row",  timeit([&] { k_write_row<<<nb, BS>>>(n, C, s); }, rep), rowMB + sMB };
  r[k++] = { "write_lds",  timeit([&] { k_write_lds<<<nb, BS>>>(n, C, s); }, rep), rowMB + sMB };
  r[k++] = { "axpy_row",   timeit([&] { k_axpy_row<<<nb, BS>>>(n, A, B, C, 0.75, 0.25); }, rep), 3 * rowMB };
  r[k++] = { "axpy_lds",   timeit([&] { k_axpy_lds<<<nb, BS>>>(n, A, B, C, 0.75, 0.25); }, rep), 3 * rowMB };
  const size_t n2 = nd / 2;
  r[k++] = { "axpy_flat",  timeit([&] { k_axpy_flat<<<(unsigned)((n2 + BS - 1) / BS), BS>>>(n2, (const double2*)A, (const double2*)B, (double2*)C, 0.75, 0.25); }, rep), 3 * rowMB };
  printf("{\"rows\": %d, \"row_bytes\": %d", n, NP * 8);
  for (int i = 0; i < k; ++i) printf(", \"%s\": {\"ms\": %.4f, \"TBps\": %.3f}", r[i].name, r[i].ms, r[i].MB / r[i].ms / 1e3);
  printf("}\n");
  return 0;
}
