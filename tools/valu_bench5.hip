// Micro-benchmark 5: one 64-bit add (v_lshl_add_u64) producing two 32-bit sums, against two
// v_add_u32, each feeding v_max3_i32 in the enumeration kernel's order.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;
template <int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters, int seed) {
  // f pairs: (f0,f1), (f2,f3) as 64-bit; g duplicated in both halves
  u64 fa = ((u64)(seed + threadIdx.x) << 32) | (unsigned)(seed * 3 + threadIdx.x);
  u64 fb = ((u64)(seed * 5 + threadIdx.x) << 32) | (unsigned)(seed * 7 + threadIdx.x);
  int a0 = -1 << 30, a1 = a0, a2 = a0, a3 = a0;
  u64 g0 = ((u64)seed << 32) | (unsigned)seed, g1 = g0 + 0x300000003ull, g2 = g0 + 0x500000005ull, g3 = g0 + 0x700000007ull;
  for (int it = 0; it < iters; ++it) {
    u64 t0, t1;
#define PAIR(F, GA, GB, ACC_LO, ACC_HI)                                                         \
    asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(t0) : "v"(F), "v"(GA));                    \
    asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(t1) : "v"(F), "v"(GB));                    \
    asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(ACC_LO) : "v"((int)t0), "v"((int)t1));        \
    asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(ACC_HI) : "v"((int)(t0 >> 32)), "v"((int)(t1 >> 32)));
    if (MODE == 0) {
      PAIR(fa, g0, g1, a0, a1) PAIR(fb, g0, g1, a2, a3) PAIR(fa, g2, g3, a0, a1) PAIR(fb, g2, g3, a2, a3)
    } else {
      int f0 = (int)fa, f1 = (int)(fa >> 32), f2 = (int)fb, f3 = (int)(fb >> 32);
      int x0, x1;
#define TRI(F, GA, GB, ACC)                                                    \
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(x0) : "v"(F), "v"((int)GA));     \
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(x1) : "v"(F), "v"((int)GB));     \
      asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(ACC) : "v"(x0), "v"(x1));
      TRI(f0, g0, g1, a0) TRI(f1, g0, g1, a1) TRI(f2, g0, g1, a2) TRI(f3, g0, g1, a3)
      TRI(f0, g2, g3, a0) TRI(f1, g2, g3, a1) TRI(f2, g2, g3, a2) TRI(f3, g2, g3, a3)
    }
    g0 += 0x100000001ull; g1 += 0x200000002ull; g2 += 0x300000003ull; g3 += 0x500000005ull;
  }
  int s = a0 ^ a1 ^ a2 ^ a3;
  if (s == 0x12345678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
int run(const char *name, int *d_out) {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  printf("%-36s", name);
  for (int wps : {2, 4, 6, 8}) {
    int blocks = cus * wps, iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 100, 1);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("  w/SIMD=%d: %.2f", wps, 2.4e9 * (ms / 1e3) / ((double)iters * 8 * wps));
  }
  printf("   cycles@2.4GHz per 2 candidates x 64 lanes\n");
  return 0;
}
int main() {
  int *d_out;
  CHK(hipMalloc((void **)&d_out, 256 * 8 * 256 * sizeof(int)));
  run<0>("v_lshl_add_u64 pairs + v_max3_i32", d_out);
  run<1>("v_add_u32 + v_max3_i32", d_out);
  return 0;
}
