// Micro-benchmark 11: issue rate of the maximum-type VALU opcodes gfx950 has beside v_max3_f32 with 16 independent accumulators per
// lane (no back-to-back dependent instructions), every CU busy, 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int N = 16;
#define OP2(name) for (int r = 0; r < N; ++r) asm volatile(name " %0, %0, %1" : "+v"(a[r]) : "v"(g));
#define OP3(name) for (int r = 0; r < N; ++r) asm volatile(name " %0, %0, %1, %2" : "+v"(a[r]) : "v"(g), "v"(h));
template <int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters, int seed) {
  int a[N];
#pragma unroll
  for (int r = 0; r < N; ++r) a[r] = seed * (r + 1) + threadIdx.x;
  int g = seed + 11, h = seed + 17;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      if (MODE == 0) { _Pragma("unroll") OP3("v_max3_f32") }
      if (MODE == 1) { _Pragma("unroll") OP3("v_maximum3_f32") }
      if (MODE == 2) { _Pragma("unroll") OP3("v_minimum3_f32") }
      if (MODE == 3) { _Pragma("unroll") OP3("v_pk_maximum3_f16") }
      if (MODE == 4) { _Pragma("unroll") OP2("v_pk_max_f16") }
      if (MODE == 5) { _Pragma("unroll") OP3("v_max3_f16") }
      if (MODE == 6) { _Pragma("unroll") OP3("v_max3_i16") }
      if (MODE == 7) { _Pragma("unroll") OP2("v_pk_max_u16") }
      if (MODE == 8) { _Pragma("unroll") OP2("v_max_u16") }
      if (MODE == 9) { _Pragma("unroll") OP2("v_add_f32") }
      if (MODE == 10) { _Pragma("unroll") OP2("v_mul_f32") }
      if (MODE == 11) { _Pragma("unroll") OP3("v_min3_f32") }
      if (MODE == 12) { _Pragma("unroll") OP3("v_med3_f32") }
      if (MODE == 13) { _Pragma("unroll") OP3("v_max3_u32") }
      if (MODE == 14) { _Pragma("unroll") OP2("v_pk_min_i16") }
      if (MODE == 15) { _Pragma("unroll") OP2("v_max_i16") }
    }
  }
  int s = 0;
#pragma unroll
  for (int r = 0; r < N; ++r) s ^= a[r];
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
  printf("%-16s", name);
  for (int wps : {1, 2, 4, 8}) {
    int blocks = cus * wps, iters = 10000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 100, 1);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    double instr_per_wave = (double)iters * 4 * N;
    printf("  w/SIMD=%d: %.2f cyc/instr", wps, 2.4e9 * (ms / 1e3) / (instr_per_wave * wps));
  }
  printf("   (cycles at 2.4 GHz per wave-instruction per SIMD)\n");
  return 0;
}
int main() {
  int *d_out;
  CHK(hipMalloc((void **)&d_out, 256 * 8 * 256 * sizeof(int)));
  run<0>("v_max3_f32", d_out);
  run<1>("v_maximum3_f32", d_out);
  run<2>("v_minimum3_f32", d_out);
  run<3>("v_pk_maximum3_f16", d_out);
  run<4>("v_pk_max_f16", d_out);
  run<5>("v_max3_f16", d_out);
  run<6>("v_max3_i16", d_out);
  run<7>("v_pk_max_u16", d_out);
  run<8>("v_max_u16", d_out);
  run<9>("v_add_f32", d_out);
  run<10>("v_mul_f32", d_out);
  run<11>("v_min3_f32", d_out);
  run<12>("v_med3_f32", d_out);
  run<13>("v_max3_u32", d_out);
  run<14>("v_pk_min_i16", d_out);
  run<15>("v_max_i16", d_out);
  return 0;
}
