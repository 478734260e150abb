// Issue rates of the integer VALU operations the rollout kernels lean on (gfx950): cycles per wave-instruction on one SIMD with 1, 2 and 4
// waves per SIMD, dependent chains of 4 independent accumulators per lane (ILP 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
#define OPS(NAME, BODY) \
__global__ __launch_bounds__(256) void k_##NAME(unsigned* out, unsigned s0, unsigned s1, unsigned long long* cyc) { \
  unsigned a0 = threadIdx.x + s0, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3; \
  unsigned long long t0 = __builtin_readcyclecounter(); \
  _Pragma("unroll 1") for (int i = 0; i < 64; ++i) { _Pragma("unroll") for (int j = 0; j < REP / 4; ++j) { BODY(a0) BODY(a1) BODY(a2) BODY(a3) } } \
  unsigned long long t1 = __builtin_readcyclecounter(); \
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3; if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0; }
#define B_XOR(x)   asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "s"(s1));
#define B_ADD(x)   asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "s"(s1));
#define B_BITOP3(x) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x) : "s"(s1), "v"(a3));
#define B_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "s"(s1));
#define B_MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "s"(s1));
#define B_MUL24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "s"(s1));
#define B_MAD64(x) { unsigned long long r_; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r_) : "v"(x), "s"(s1) : "vcc"); x = (unsigned)(r_ >> 32) ^ (unsigned)r_; }
#define B_MAD64ONLY(x) { unsigned long long r_; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r_) : "v"(x), "s"(s1) : "vcc"); x = (unsigned)(r_ >> 32); }
#define B_CVT(x)   asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x));
#define B_MULF(x)  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "s"(s1));
#define B_LSHLOR(x) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x) : "s"(s1));
#define B_MIN(x)   asm volatile("v_min_i32 %0, %0, %1" : "+v"(x) : "s"(s1));
#define B_PKADD(x) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "s"(s1));
#define B_PKMAX(x) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x) : "s"(s1));
#define B_MADU24(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "s"(s1));
#define B_ADD3(x) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x) : "s"(s1));
OPS(xor, B_XOR) OPS(add, B_ADD) OPS(bitop3, B_BITOP3) OPS(mullo, B_MULLO) OPS(mulhi, B_MULHI) OPS(mul24, B_MUL24) OPS(mad64xor, B_MAD64) OPS(mad64, B_MAD64ONLY)
OPS(cvt, B_CVT) OPS(mulf, B_MULF) OPS(lshlor, B_LSHLOR) OPS(min, B_MIN) OPS(pkadd, B_PKADD) OPS(pkmax, B_PKMAX) OPS(madu24, B_MADU24) OPS(add3, B_ADD3)
template <typename K> void run(const char* name, K kern, int extra) {
  unsigned* out; unsigned long long* cyc; hipMalloc(&out, 256 * 4096 * 4); hipMalloc(&cyc, 4096 * 8);
  for (int wps : {1, 2, 4, 8}) {               // waves per SIMD: blocks of 256 threads = 1 wave per SIMD; wps blocks per CU
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1u, 12345u, cyc);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1u, 12345u, cyc);
    hipDeviceSynchronize();
    unsigned long long h[4096]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; ++i) s += (double)h[i];
    const double per = s / blocks / (64.0 * REP);        // cycles per instruction-slot of one wave
    printf("%-10s %d waves/SIMD: %6.2f cycles per op per wave -> %5.2f cycles of SIMD time per wave-instruction%s\n", name, wps, per, per / wps, extra ? "  (+1 xor per op)" : "");
  }
  hipFree(out); hipFree(cyc);
}
int main() {
  run("xor", k_xor, 0); run("add", k_add, 0); run("bitop3", k_bitop3, 0); run("mul_lo", k_mullo, 0); run("mul_hi", k_mulhi, 0); run("mul_u24", k_mul24, 0);
  run("mad_u64", k_mad64, 0); run("mad64+xor", k_mad64xor, 1); run("cvt_f32", k_cvt, 0); run("mul_f32", k_mulf, 0); run("lshl_or", k_lshlor, 0); run("min_i32", k_min, 0);
  run("pk_add16", k_pkadd, 0); run("pk_max16", k_pkmax, 0); run("mad_u24", k_madu24, 0); run("add3", k_add3, 0);
  return 0;
}
