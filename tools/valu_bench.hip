// Micro-benchmark: sustained rate of the integer VALU instructions the enumeration
// kernel is made of (v_add_u32, v_max_i32, v_max3_i32) on every CU of the chip, at
// 1, 2, 4 and 8 waves per SIMD.  Prints lane-ops per second per variant.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_bench.hip -o tools/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int R = 8;

template <int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters, int seed) {
  int f[R], acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { f[r] = seed * (r + 1) + threadIdx.x; acc[r] = -(1 << 30); }
  int g0 = seed, g1 = seed + 3, g2 = seed + 5, g3 = seed + 7;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (MODE == 0) {  // add + max per candidate
        acc[r] = max(acc[r], f[r] + g0); acc[r] = max(acc[r], f[r] + g1);
        acc[r] = max(acc[r], f[r] + g2); acc[r] = max(acc[r], f[r] + g3);
        asm volatile("" : "+v"(acc[r]));
      } else if (MODE == 1) {  // 2 adds + max3 per 2 candidates
        int t0 = f[r] + g0, t1 = f[r] + g1, t2 = f[r] + g2, t3 = f[r] + g3;
        int a;
        asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(a) : "v"(acc[r]), "v"(t0), "v"(t1));
        asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(acc[r]) : "v"(a), "v"(t2), "v"(t3));
      } else if (MODE == 2) {  // adds only (4 per r)
        int t0, t1, t2, t3;
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(t0) : "v"(f[r]), "v"(g0));
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(t1) : "v"(f[r]), "v"(g1));
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(t2) : "v"(f[r]), "v"(g2));
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(t3) : "v"(f[r]), "v"(g3));
        acc[r] ^= t0 ^ t1 ^ t2 ^ t3;  // 4 more ops; counted below
      } else if (MODE == 3) {  // max only (4 per r)
        int a = acc[r];
        asm volatile("v_max_i32 %0, %0, %1" : "+v"(a) : "v"(g0));
        asm volatile("v_max_i32 %0, %0, %1" : "+v"(a) : "v"(g1));
        asm volatile("v_max_i32 %0, %0, %1" : "+v"(a) : "v"(g2));
        asm volatile("v_max_i32 %0, %0, %1" : "+v"(a) : "v"(g3));
        acc[r] = a;
      } else if (MODE == 5) {  // v_max_f32 on integer bit patterns (4 per r)
        int a = acc[r];
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(g0));
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(g1));
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(g2));
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(g3));
        acc[r] = a;
      } else if (MODE == 6) {  // 2 x v_add_u32 + v_max3_f32 per 2 candidates
        int t0 = f[r] + g0, t1 = f[r] + g1, t2 = f[r] + g2, t3 = f[r] + g3;
        int a;
        asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(a) : "v"(acc[r]), "v"(t0), "v"(t1));
        asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(acc[r]) : "v"(a), "v"(t2), "v"(t3));
      } else if (MODE == 7) {  // v_add_u32 + v_max_f32 per candidate
        int t0 = f[r] + g0, t1 = f[r] + g1, t2 = f[r] + g2, t3 = f[r] + g3;
        int a = acc[r];
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(t0));
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(t1));
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(t2));
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(t3));
        acc[r] = a;
      } else if (MODE == 8) {  // v_max3_f32 only (2 per r)
        int a = acc[r];
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(g0), "v"(g1));
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(g2), "v"(g3));
        acc[r] = a;
      } else if (MODE == 9) {  // v_max_u32 (4 per r)
        int a = acc[r];
        asm volatile("v_max_u32 %0, %0, %1" : "+v"(a) : "v"(g0));
        asm volatile("v_max_u32 %0, %0, %1" : "+v"(a) : "v"(g1));
        asm volatile("v_max_u32 %0, %0, %1" : "+v"(a) : "v"(g2));
        asm volatile("v_max_u32 %0, %0, %1" : "+v"(a) : "v"(g3));
        acc[r] = a;
      } else if (MODE == 10) {  // v_pk_add_f32-free check: v_add_f32 + v_max_f32
        int a = acc[r], t0, t1;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(t0) : "v"(f[r]), "v"(g0));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(t1) : "v"(f[r]), "v"(g1));
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(t0), "v"(t1));
        acc[r] = a;
      } else if (MODE == 4) {  // packed 16-bit: v_pk_add_i16 + v_pk_max_i16, two candidates per op pair
        int t0, t1, a = acc[r];
        asm volatile("v_pk_add_i16 %0, %1, %2" : "=v"(t0) : "v"(f[r]), "v"(g0));
        asm volatile("v_pk_add_i16 %0, %1, %2" : "=v"(t1) : "v"(f[r]), "v"(g1));
        asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a) : "v"(t0));
        asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a) : "v"(t1));
        acc[r] = a;
      }
    }
    g0 += 1; g1 += 2; g2 += 3; g3 += 5;
  }
  int s = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) s ^= acc[r];
  if (s == 0x12345678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
int run(const char *name, double ops_per_iter_per_lane, int *d_out) {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD
    int blocks = cus * wps;
    int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 100, 1);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    double lane_ops = (double)blocks * 256 * iters * ops_per_iter_per_lane;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f Tlane-op/s  (%.2f cycles@2.4GHz per wave-instr per SIMD)\n", name, wps, ms,
           lane_ops / ms / 1e9, 2.4e9 * (ms / 1e3) / ((double)iters * ops_per_iter_per_lane * wps));
  }
  return 0;
}

int main() {
  int *d_out;
  CHK(hipMalloc((void **)&d_out, 256 * 8 * 256 * 4 * sizeof(int)));
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  if (run<0>("add+max (2 ops/cand)", R * 8.0 + 4, d_out)) return 1;
  if (run<1>("2add+max3 (1.5 ops/cand)", R * 6.0 + 4, d_out)) return 1;
  if (run<2>("v_add_u32 + xor", R * 8.0 + 4, d_out)) return 1;
  if (run<3>("v_max_i32", R * 4.0 + 4, d_out)) return 1;
  if (run<4>("v_pk_add_i16+v_pk_max_i16", R * 4.0 + 4, d_out)) return 1;
  if (run<5>("v_max_f32", R * 4.0 + 4, d_out)) return 1;
  if (run<6>("2 v_add_u32 + v_max3_f32", R * 6.0 + 4, d_out)) return 1;
  if (run<7>("v_add_u32 + v_max_f32", R * 8.0 + 4, d_out)) return 1;
  if (run<8>("v_max3_f32", R * 2.0 + 4, d_out)) return 1;
  if (run<9>("v_max_u32", R * 4.0 + 4, d_out)) return 1;
  if (run<10>("2 v_add_f32 + v_max3_f32", R * 3.0 + 4, d_out)) return 1;
  return 0;
}
