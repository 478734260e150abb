// ub_dppscan.hip -- check the DPP wave64 inclusive scan used by block_exscan (phx_generic.hip) against a serial scan,
// and time it against the __shfl_up (ds_bpermute) version.   hipcc --offload-arch=gfx950 -O3 ub_dppscan.hip -o ub_dppscan
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ int wave_incl_scan_dpp(int v) {
  int x = v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return x;
}
__device__ __forceinline__ int wave_incl_scan_shfl(int v) {
  int incl = v;
  for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if ((threadIdx.x & 63) >= off) incl += t; }
  return incl;
}
__global__ void k_check(const int* in, int* out_dpp, int* out_shfl) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  out_dpp[i] = wave_incl_scan_dpp(in[i]);
  out_shfl[i] = wave_incl_scan_shfl(in[i]);
}
template <bool DPP>
__global__ void k_time(const int* in, int* out, int reps) {
  int v = in[threadIdx.x];
  for (int r = 0; r < reps; ++r) v = (DPP ? wave_incl_scan_dpp(v) : wave_incl_scan_shfl(v)) & 0xffff;
  out[blockIdx.x * 64 + threadIdx.x] = v;
}
int main() {
  const int n = 64 * 64;
  std::vector<int> h(n), a(n), b(n);
  for (int i = 0; i < n; ++i) h[i] = (i * 2654435761u >> 20) % 97;
  int *d, *o1, *o2; hipMalloc(&d, n * 4); hipMalloc(&o1, n * 4); hipMalloc(&o2, n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_check, dim3(64), dim3(64), 0, 0, d, o1, o2);
  hipMemcpy(a.data(), o1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 64; ++w) { int run = 0; for (int l = 0; l < 64; ++l) { run += h[w * 64 + l]; if (a[w * 64 + l] != run || b[w * 64 + l] != run) ++bad; } }
  printf("dpp scan vs serial: %d mismatches\n", bad);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int dpp = 0; dpp < 2; ++dpp) {
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (dpp) hipLaunchKernelGGL(k_time<true>, dim3(1), dim3(64), 0, 0, d, o1, 10000);
      else hipLaunchKernelGGL(k_time<false>, dim3(1), dim3(64), 0, 0, d, o1, 10000);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%s: %.1f ns per wave scan (one wave, dependent chain)\n", dpp ? "dpp " : "shfl", best * 1e6 / 10000);
  }
  return bad != 0;
}
