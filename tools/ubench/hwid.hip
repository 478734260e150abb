// where do the waves of a 256-thread workgroup land?  (SIMD / CU / XCC per wave)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256) void k(unsigned* out, int spin) {
  extern __shared__ char sm[];
  unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
  volatile float x = 1.f;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;     // keep the WG alive so that all are co-resident
  if (threadIdx.x == 300) sm[0] = (char)x;
  if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}
int main() {
  const int nb = 1024;
  unsigned* d; hipMalloc(&d, nb * 4 * 2 * 4);
  hipLaunchKernelGGL(k, dim3(nb), dim3(256), 30 * 1024, 0, d, 20000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nb * 8); hipMemcpy(h.data(), d, nb * 32, hipMemcpyDeviceToHost);
  int hist[4][4] = {{0}};           // [wave index][simd]
  std::map<unsigned, int> percu;     // (xcc, se, sh, cu) -> WGs
  std::map<unsigned, std::vector<int>> cu_wave0_simd;
  for (int b = 0; b < nb; ++b) {
    for (int w = 0; w < 4; ++w) { unsigned hw = h[(b * 4 + w) * 2]; hist[w][(hw >> 4) & 3]++; }
    unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 15;
    unsigned key = (xcc << 16) | (hw & 0xff00);
    percu[key]++; cu_wave0_simd[key].push_back((hw >> 4) & 3);
  }
  for (int w = 0; w < 4; ++w) printf("wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  std::map<int, int> wgs; for (auto& kv : percu) wgs[kv.second]++;
  for (auto& kv : wgs) printf("CUs with %d WGs: %d\n", kv.first, kv.second);
  int shown = 0;
  for (auto& kv : cu_wave0_simd) { if (shown++ >= 6) break; printf("cu %06x wave0 simds:", kv.first); for (int s : kv.second) printf(" %d", s); printf("\n"); }
  for (int b = 0; b < 20; ++b) { printf("WG %d:", b); for (int w = 0; w < 4; ++w) { unsigned hw = h[(b * 4 + w) * 2]; printf(" [xcc%u se%u cu%u simd%u wv%u]", h[(b*4+w)*2+1] & 15, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15); } printf("\n"); }
  return 0;
}
