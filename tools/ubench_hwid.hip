// Where does the dispatcher put the waves of a 4-wave workgroup that owns half a CU's LDS?
// (gfx950; the DG-P1 tile kernel's shape: 256 lanes, 79 KB of LDS, two workgroups per CU.)
// Every wave records HW_REG_HW_ID (wave slot, SIMD, CU, SE, workgroup slot) and its start / end
// time; the host prints, per wave index, the histogram of SIMD ids, and how often the two
// workgroups resident on one CU have the same / different TG_ID parity.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_hwid.hip -o tools/ubench_hwid && tools/ubench_hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>

__global__ __launch_bounds__(256, 2) void k(unsigned* hw, unsigned long long* tm, int spin)
{
  __shared__ double big[79360 / 8];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // first instruction of the wave
  const int tid = threadIdx.x;
  big[tid] = tid;
  __syncthreads();
  double x = big[(tid * 7) & 255];
  for (int i = 0; i < spin; ++i) x = fma(x, 1.0000001, 1e-9);
  big[tid] = x;
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if ((tid & 63) == 0) {
    const unsigned id = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID[3:0]
    const int w = blockIdx.x * 4 + (tid >> 6);
    hw[w] = (id & 0x0fffffffu) | (xcc << 28);
    tm[2 * w] = t0;
    tm[2 * w + 1] = t1 + (unsigned long long)(big[0] * 1e-300);
  }
}

int main()
{
  const int nb = 4096;
  unsigned* dhw; unsigned long long* dtm;
  hipMalloc(&dhw, nb * 4 * sizeof(unsigned));
  hipMalloc(&dtm, nb * 8 * sizeof(unsigned long long));
  k<<<nb, 256>>>(dhw, dtm, 20000);
  hipDeviceSynchronize();
  std::vector<unsigned> hw(nb * 4);
  std::vector<unsigned long long> tm(nb * 8);
  hipMemcpy(hw.data(), dhw, hw.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  hipMemcpy(tm.data(), dtm, tm.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  // gfx9 HW_ID: WAVE_ID[3:0] SIMD_ID[5:4] PIPE_ID[7:6] CU_ID[11:8] SH_ID[12] SE_ID[15:13] TG_ID[19:16] ...
  int simd_of_wave[4][4] = {};
  std::map<unsigned, int> tg_hist;
  for (int b = 0; b < nb; ++b)
    for (int w = 0; w < 4; ++w) {
      const unsigned id = hw[b * 4 + w];
      simd_of_wave[w][(id >> 4) & 3]++;
      if (w == 0) tg_hist[(id >> 16) & 15]++;
    }
  printf("wave index -> SIMD id histogram over %d workgroups\n", nb);
  for (int w = 0; w < 4; ++w)
    printf("  wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, simd_of_wave[w][0], simd_of_wave[w][1],
           simd_of_wave[w][2], simd_of_wave[w][3]);
  printf("TG_ID histogram (wave 0):");
  for (auto& kv : tg_hist) printf("  %u:%d", kv.first, kv.second);
  printf("\n");
  // all waves of one workgroup on distinct SIMDs?
  int distinct = 0, same_rot = 0;
  for (int b = 0; b < nb; ++b) {
    int s[4];
    for (int w = 0; w < 4; ++w) s[w] = (hw[b * 4 + w] >> 4) & 3;
    int mask = 0;
    for (int w = 0; w < 4; ++w) mask |= 1 << s[w];
    if (mask == 15) ++distinct;
    if (s[1] == ((s[0] + 1) & 3) && s[2] == ((s[0] + 2) & 3) && s[3] == ((s[0] + 3) & 3)) ++same_rot;
  }
  printf("workgroups with their 4 waves on 4 distinct SIMDs: %d of %d; in rotation order: %d\n", distinct, nb, same_rot);
  // pairs of workgroups that overlap in time on one CU: wave-0 SIMD equal? TG_ID parity equal?
  struct W { unsigned cu; unsigned long long t0, t1; int s0, tg; };
  std::vector<W> ws(nb);
  for (int b = 0; b < nb; ++b) {
    const unsigned id = hw[b * 4];
    // xcc id is not in HW_ID on gfx9; use (SE, SH, CU) + XCC_ID register is separate -- group by time overlap only within equal (se,sh,cu)
    ws[b] = { ((id >> 8) & 0xff) | ((id >> 28) << 8), tm[8 * b], tm[8 * b + 1], (int)((id >> 4) & 3), (int)((id >> 16) & 15) };
  }
  long long pairs = 0, same_s0 = 0, same_tgpar = 0;
  for (int a = 0; a < nb; ++a)
    for (int b = a + 1; b < nb; ++b)
      if (ws[a].cu == ws[b].cu && ws[a].t0 < ws[b].t1 && ws[b].t0 < ws[a].t1) {
        ++pairs;
        if (ws[a].s0 == ws[b].s0) ++same_s0;
        if ((ws[a].tg & 1) == (ws[b].tg & 1)) ++same_tgpar;
      }
  printf("time-overlapping pairs on one (XCC,SE,SH,CU): %lld; wave 0 on the same SIMD: %lld; same TG_ID parity: %lld\n",
         pairs, same_s0, same_tgpar);
  // dispatch gap: on one (XCC, SE, SH, CU, workgroup slot), the time from the end of a workgroup (after its last
  // barrier) to the first instruction of the next one; s_memrealtime ticks at 100 MHz
  std::map<unsigned, std::vector<std::pair<unsigned long long, unsigned long long>>> slots;
  for (int b = 0; b < nb; ++b) slots[(ws[b].cu << 4) | (unsigned)ws[b].tg].push_back({ ws[b].t0, ws[b].t1 });
  std::vector<double> gaps, durs;
  for (auto& kv : slots) {
    auto& v = kv.second;
    std::sort(v.begin(), v.end());
    for (size_t i = 0; i + 1 < v.size(); ++i) gaps.push_back((double)((long long)v[i + 1].first - (long long)v[i].second) * 0.01);
    for (auto& p : v) durs.push_back((double)(p.second - p.first) * 0.01);
  }
  std::sort(gaps.begin(), gaps.end()); std::sort(durs.begin(), durs.end());
  if (!gaps.empty())
    printf("slots %zu; workgroup duration median %.2f us; gap end -> next start on the same slot: min %.2f median %.2f p90 %.2f max %.2f us (%zu gaps)\n",
           slots.size(), durs[durs.size() / 2], gaps.front(), gaps[gaps.size() / 2], gaps[gaps.size() * 9 / 10], gaps.back(), gaps.size());
  return 0;
}
