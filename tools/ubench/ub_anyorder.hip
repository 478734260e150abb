// Can launch n + 1 of a chain of DEPENDENT launches start on a CU as soon as launch n's workgroup there has left, instead of after
// the whole of launch n (last workgroup + end-of-kernel write-back + the dispatcher's gap)?
//   mode 0: hipLaunchKernelGGL back to back (the AQL barrier bit orders the launches);
//   mode 1: hipExtLaunchKernelGGL(.., hipExtAnyOrderLaunch): the packets of one queue are still DISPATCHED in order, but packet n + 1
//           no longer waits for packet n to complete; workgroup w of launch n + 1 waits for workgroup w of launch n through a
//           sequence word in HBM (release store / acquire spin, agent scope; bounded by wall clock).
// Every workgroup takes 160 KB of LDS (one per CU, like phx_sc_rollout_sw_kernel), "works" for a time that varies by workgroup and
// launch (s_sleep loop, or a stream of stores), and stamps entry / go / exit with the 100 MHz wall clock.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ub_anyorder.hip -o /tmp/ub_anyorder && /tmp/ub_anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Args { unsigned long long* stamps; int* seq; int* err; float4* sink; int launch, chained, check, work_us, grid, stores, spread; };

__global__ __launch_bounds__(1024) void k(const Args a) {
  extern __shared__ char smem[];
  const int w = blockIdx.x, tid = threadIdx.x;
  const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) {
    if (a.chained) {
      while (__hip_atomic_load(&a.seq[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < a.launch) {
        __builtin_amdgcn_s_sleep(8);
        if (__builtin_amdgcn_s_memrealtime() - t_entry > 2000000ull) { atomicOr(a.err, 1); break; }     // 20 ms: never hang the box
      }
    } else if (a.check && a.seq[w] != a.launch) atomicOr(a.err, 2);
    smem[0] = 1;
  }
  __syncthreads();
  const unsigned long long t_go = __builtin_amdgcn_s_memrealtime();
  // the work: work_us -+ spread / 2 percent, by (workgroup, launch)
  const unsigned h = ((unsigned)w * 2654435761u + (unsigned)a.launch * 40503u) >> 24;      // 0..255
  const unsigned long long dur = (unsigned long long)a.work_us * (100 - a.spread / 2 + (h * a.spread) / 256);    // in 10 ns ticks (x 100 / 100)
  if (a.stores) {
    float4* p = a.sink + ((size_t)(a.launch & 1) * a.grid + w) * (size_t)(1 << 16);      // 1 MB per workgroup and launch parity
    int i = tid;
    while (__builtin_amdgcn_s_memrealtime() - t_go < dur) { __builtin_nontemporal_store(1.0f, (float*)&p[i & 0xffff]); i += 1024; }
  } else {
    while (__builtin_amdgcn_s_memrealtime() - t_go < dur) __builtin_amdgcn_s_sleep(16);
  }
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store(&a.seq[w], a.launch + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long* s = a.stamps + ((size_t)a.launch * a.grid + w) * 4;
    unsigned id, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    s[0] = t_entry; s[1] = t_go; s[2] = __builtin_amdgcn_s_memrealtime(); s[3] = ((unsigned long long)(xcc & 15u) << 16) | ((id >> 8) & 0xffu);   // XCC | se, sh, cu
  }
}

int main(int argc, char** argv) {
  const int n_launch = 60, lds = 160 * 1024 - 512;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipStream_t st; hipStreamCreate(&st);
  for (int spread : {0, 4, 40}) for (int grid : {256, 512}) for (int stores : {0, 1}) for (int work_us : {15}) for (int mode : {0, 1, 2}) {
    Args a;
    hipMalloc(&a.stamps, (size_t)n_launch * grid * 4 * 8); hipMalloc(&a.seq, grid * 4); hipMalloc(&a.err, 4);
    hipMalloc(&a.sink, (size_t)2 * grid << 20);
    hipMemset(a.seq, 0, grid * 4); hipMemset(a.err, 0, 4); hipMemset(a.stamps, 0, (size_t)n_launch * grid * 4 * 8);
    a.chained = mode == 1; a.check = mode == 0; a.work_us = work_us; a.spread = spread; a.grid = grid; a.stores = stores;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int l = 0; l < n_launch; ++l) {
      a.launch = l;
      if (mode == 0 || l == 0) hipLaunchKernelGGL(k, dim3(grid), dim3(1024), lds, st, a);
      else hipExtLaunchKernelGGL(k, dim3(grid), dim3(1024), lds, st, nullptr, nullptr, hipExtAnyOrderLaunch, a);
    }
    hipEventRecord(e1, st);
    hipError_t e = hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)n_launch * grid * 4); hipMemcpy(h.data(), a.stamps, h.size() * 8, hipMemcpyDeviceToHost);
    int err = 0; hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost);
    // overlap: workgroups of launch l that ENTERED before the last workgroup of launch l - 1 left; mean gap between launches
    long long early = 0; double gap = 0, wait = 0;
    for (int l = 1; l < n_launch; ++l) {
      unsigned long long last_exit = 0, first_entry = ~0ull;
      for (int w = 0; w < grid; ++w) last_exit = std::max(last_exit, h[((size_t)(l - 1) * grid + w) * 4 + 2]);
      for (int w = 0; w < grid; ++w) { const unsigned long long en = h[((size_t)l * grid + w) * 4]; first_entry = std::min(first_entry, en); if (en < last_exit) ++early;
        wait += (double)(h[((size_t)l * grid + w) * 4 + 1] - en) * 0.01; }
      gap += (double)(long long)(first_entry - last_exit) * 0.01;
    }
    // per-CU timeline: idle time between a workgroup's exit and the next workgroup's entry on the same CU
    double idle = 0, busy = 0; long long n_idle = 0; int n_cu = 0;
    { std::vector<std::vector<std::pair<unsigned long long, unsigned long long>>> cu(16 << 16);
      for (size_t i = 0; i < (size_t)n_launch * grid; ++i) cu[h[i * 4 + 3] & 0xfffff].push_back({h[i * 4], h[i * 4 + 2]});
      for (auto& v : cu) { if (v.empty()) continue; ++n_cu; std::sort(v.begin(), v.end());
        for (size_t j = 0; j < v.size(); ++j) { busy += (double)(v[j].second - v[j].first) * 0.01; if (j) { idle += (double)(long long)(v[j].first - v[j - 1].second) * 0.01; ++n_idle; } } } }
    printf("  CUs seen %d | per CU: mean residency %.2f us, mean idle between consecutive workgroups %.2f us\n", n_cu, busy / ((double)n_launch * grid), idle / (double)std::max(1LL, n_idle));
    printf("spread %2d %% grid %4d %s work ~%2d us mode %d (%s): %7.2f us per launch | first entry - previous launch's last exit: mean %6.2f us | workgroups that entered early %5.1f %% | mean wait at entry %5.2f us | err %d (%s)\n",
           spread, grid, stores ? "stores" : "sleep ", work_us, mode, mode == 1 ? "any-order + handshake" : (mode == 2 ? "any-order, independent" : "barrier bit"), ms * 1e3 / n_launch, gap / (n_launch - 1),
           100.0 * early / ((double)(n_launch - 1) * grid), wait / ((double)(n_launch - 1) * grid), err, hipGetErrorString(e));
    fflush(stdout);
    hipFree(a.stamps); hipFree(a.seq); hipFree(a.err); hipFree(a.sink);
  }
  return 0;
}
