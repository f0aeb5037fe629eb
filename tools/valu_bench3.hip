// Micro-benchmark 3: instruction orderings of the enumeration inner loop
// (4 f registers, 4 running maxima, 4 broadcast g values per iteration).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define TRI(ADD, MAX3, acc, f, ga, gb, ta, tb)                                  \
  asm volatile(ADD " %0, %1, %2" : "=v"(ta) : "v"(f), "v"(ga));                 \
  asm volatile(ADD " %0, %1, %2" : "=v"(tb) : "v"(f), "v"(gb));                 \
  asm volatile(MAX3 " %0, %0, %1, %2" : "+v"(acc) : "v"(ta), "v"(tb));

template <int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters, int seed) {
  int f0 = seed + threadIdx.x, f1 = f0 * 3, f2 = f0 * 5, f3 = f0 * 7;
  int a0 = -1 << 30, a1 = a0, a2 = a0, a3 = a0;
  int g0 = seed, g1 = seed + 3, g2 = seed + 5, g3 = seed + 7;
  if (MODE == 2) { a0 = a1 = a2 = a3 = 0x00800000; f0 &= 0xfffff; f1 &= 0xfffff; f2 &= 0xfffff; f3 &= 0xfffff; }
  for (int it = 0; it < iters; ++it) {
    int t0, t1, t2, t3, t4, t5, t6, t7;
    if (MODE == 0) {  // kernel order, two shared temporaries
      TRI("v_add_u32", "v_max3_i32", a0, f0, g0, g1, t0, t1) TRI("v_add_u32", "v_max3_i32", a1, f1, g0, g1, t0, t1)
      TRI("v_add_u32", "v_max3_i32", a2, f2, g0, g1, t0, t1) TRI("v_add_u32", "v_max3_i32", a3, f3, g0, g1, t0, t1)
      TRI("v_add_u32", "v_max3_i32", a0, f0, g2, g3, t0, t1) TRI("v_add_u32", "v_max3_i32", a1, f1, g2, g3, t0, t1)
      TRI("v_add_u32", "v_max3_i32", a2, f2, g2, g3, t0, t1) TRI("v_add_u32", "v_max3_i32", a3, f3, g2, g3, t0, t1)
    } else if (MODE == 1) {  // eight adds into distinct temporaries, then four max3, twice
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(t0) : "v"(f0), "v"(g0)); asm volatile("v_add_u32 %0, %1, %2" : "=v"(t1) : "v"(f0), "v"(g1));
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(t2) : "v"(f1), "v"(g0)); asm volatile("v_add_u32 %0, %1, %2" : "=v"(t3) : "v"(f1), "v"(g1));
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(t4) : "v"(f2), "v"(g0)); asm volatile("v_add_u32 %0, %1, %2" : "=v"(t5) : "v"(f2), "v"(g1));
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(t6) : "v"(f3), "v"(g0)); asm volatile("v_add_u32 %0, %1, %2" : "=v"(t7) : "v"(f3), "v"(g1));
      asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a0) : "v"(t0), "v"(t1)); asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a1) : "v"(t2), "v"(t3));
      asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a2) : "v"(t4), "v"(t5)); asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a3) : "v"(t6), "v"(t7));
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(t0) : "v"(f0), "v"(g2)); asm volatile("v_add_u32 %0, %1, %2" : "=v"(t1) : "v"(f0), "v"(g3));
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(t2) : "v"(f1), "v"(g2)); asm volatile("v_add_u32 %0, %1, %2" : "=v"(t3) : "v"(f1), "v"(g3));
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(t4) : "v"(f2), "v"(g2)); asm volatile("v_add_u32 %0, %1, %2" : "=v"(t5) : "v"(f2), "v"(g3));
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(t6) : "v"(f3), "v"(g2)); asm volatile("v_add_u32 %0, %1, %2" : "=v"(t7) : "v"(f3), "v"(g3));
      asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a0) : "v"(t0), "v"(t1)); asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a1) : "v"(t2), "v"(t3));
      asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a2) : "v"(t4), "v"(t5)); asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a3) : "v"(t6), "v"(t7));
    } else if (MODE == 2) {  // kernel order with float opcodes on positive integer patterns
      TRI("v_add_f32", "v_max3_f32", a0, f0, g0, g1, t0, t1) TRI("v_add_f32", "v_max3_f32", a1, f1, g0, g1, t0, t1)
      TRI("v_add_f32", "v_max3_f32", a2, f2, g0, g1, t0, t1) TRI("v_add_f32", "v_max3_f32", a3, f3, g0, g1, t0, t1)
      TRI("v_add_f32", "v_max3_f32", a0, f0, g2, g3, t0, t1) TRI("v_add_f32", "v_max3_f32", a1, f1, g2, g3, t0, t1)
      TRI("v_add_f32", "v_max3_f32", a2, f2, g2, g3, t0, t1) TRI("v_add_f32", "v_max3_f32", a3, f3, g2, g3, t0, t1)
    } else if (MODE == 3) {  // integer add, float max3
      TRI("v_add_u32", "v_max3_f32", a0, f0, g0, g1, t0, t1) TRI("v_add_u32", "v_max3_f32", a1, f1, g0, g1, t0, t1)
      TRI("v_add_u32", "v_max3_f32", a2, f2, g0, g1, t0, t1) TRI("v_add_u32", "v_max3_f32", a3, f3, g0, g1, t0, t1)
      TRI("v_add_u32", "v_max3_f32", a0, f0, g2, g3, t0, t1) TRI("v_add_u32", "v_max3_f32", a1, f1, g2, g3, t0, t1)
      TRI("v_add_u32", "v_max3_f32", a2, f2, g2, g3, t0, t1) TRI("v_add_u32", "v_max3_f32", a3, f3, g2, g3, t0, t1)
    } else if (MODE == 4) {  // float add, integer max3
      TRI("v_add_f32", "v_max3_i32", a0, f0, g0, g1, t0, t1) TRI("v_add_f32", "v_max3_i32", a1, f1, g0, g1, t0, t1)
      TRI("v_add_f32", "v_max3_i32", a2, f2, g0, g1, t0, t1) TRI("v_add_f32", "v_max3_i32", a3, f3, g0, g1, t0, t1)
      TRI("v_add_f32", "v_max3_i32", a0, f0, g2, g3, t0, t1) TRI("v_add_f32", "v_max3_i32", a1, f1, g2, g3, t0, t1)
      TRI("v_add_f32", "v_max3_i32", a2, f2, g2, g3, t0, t1) TRI("v_add_f32", "v_max3_i32", a3, f3, g2, g3, t0, t1)
    }
    g0 += 1; g1 += 2; g2 += 3; g3 += 5;
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
  printf("%-44s", name);
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
  printf("   cycles@2.4GHz per (2 add + max3) per SIMD\n");
  return 0;
}
int main() {
  int *d_out;
  CHK(hipMalloc((void **)&d_out, 256 * 8 * 256 * sizeof(int)));
  run<0>("i32: add,add,max3 per register (kernel order)", d_out);
  run<1>("i32: 8 adds then 4 max3", d_out);
  run<2>("f32 opcodes, kernel order", d_out);
  run<3>("v_add_u32 + v_max3_f32, kernel order", d_out);
  run<4>("v_add_f32 + v_max3_i32, kernel order", d_out);
  return 0;
}
