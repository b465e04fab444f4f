// fp64 VALU micro-benchmark for gfx950: cycles per wave-instruction of
// v_fma_f64 / v_mul_f64 / v_add_f64 / v_rcp_f64 / v_rsq_f64 / v_cndmask_b32 for
// 1..8 independent chains per wave and 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP, int ILP>
__global__ void k(double* out, int iters, double a0, double b0)
{
  double x[ILP];
#pragma unroll
  for (int j = 0; j < ILP; ++j) x[j] = a0 + threadIdx.x * 1e-9 + j;
  const double b = b0, c = 1.0 - b0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int j = 0; j < ILP; ++j) {
        if (OP == 0) x[j] = fma(x[j], b, c);
        else if (OP == 1) x[j] = x[j] * b;
        else if (OP == 2) x[j] = x[j] + c;
        else if (OP == 3) x[j] = __builtin_amdgcn_rcp(x[j]);
        else if (OP == 4) x[j] = __builtin_amdgcn_rsq(x[j]);
        else if (OP == 5) x[j] = (x[j] > 1.5) ? b : x[j] + 0.0 * c;   // cmp + cndmask pair
        else if (OP == 6) x[j] = fmin(x[j], b) + c;
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int j = 0; j < ILP; ++j) s += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (double)(t1 - t0) * 1e-300;
  if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[0] = t1 - t0;
}

template <int OP, int ILP> double run(int waves_per_simd, double* d, int iters)
{
  // one workgroup per CU sized waves_per_simd*4 waves; returns ns per
  // wave-instruction per SIMD from the wall clock (hipEvents)
  const int threads = 64 * 4 * waves_per_simd;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<OP, ILP>), dim3(256), dim3(threads), 0, 0, d, iters, 1.0000001, 0.9999999);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<OP, ILP>), dim3(256), dim3(threads), 0, 0, d, iters, 1.0000001, 0.9999999);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  const double ninstr_per_simd = (double)iters * 16 * ILP * waves_per_simd;
  return ms * 1e6 / ninstr_per_simd;   // ns per wave-instruction per SIMD
}

int main()
{
  double* d;
  hipMalloc(&d, 256 * 1024 * 8);
  const char* names[] = { "fma_f64", "mul_f64", "add_f64", "rcp_f64", "rsq_f64", "cmp+cnd", "min+add" };
  printf("ns per wave-instruction per SIMD (wall clock); x2.4 = cycles at 2.4 GHz\n");
  printf("%-8s %5s %8s %8s %8s\n", "op", "ILP", "1w/SIMD", "2w/SIMD", "4w/SIMD");
#define ROW(OP, ILP) { printf("%-8s %5d", names[OP], ILP); for (int w : {1, 2, 4}) printf(" %8.2f", run<OP, ILP>(w, d, 20000)); printf("\n"); }
  ROW(0, 1) ROW(0, 2) ROW(0, 4) ROW(0, 8)
  ROW(1, 1) ROW(1, 4) ROW(2, 1) ROW(2, 4)
  ROW(3, 1) ROW(3, 4) ROW(4, 1) ROW(4, 4)
  ROW(5, 1) ROW(5, 4) ROW(6, 1) ROW(6, 4)
  return 0;
}
