// Micro-benchmark 6: float32 forms of the candidate loop's inner step (two sums f + g and one
// max3 per two candidates) with g as scalar operands, as the kernel has them:
//   0  2 x v_add_f32 (scalar g) + v_max3_f32, kernel order (sums consumed two blocks later)
//   1  2 x v_add_f32 (scalar g) + v_max3_f32, each max3 right after its two adds
//   2  v_pk_add_f32 (scalar pair g[i], g[i+1]; f broadcast by op_sel) + v_max3_f32, kernel order
//   3  v_pk_add_f32 alone
//   4  v_add_f32 alone (scalar g)
//   5  v_max3_f32 alone
// Only the rate matters; results are meaningless.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_bench6.hip -o tools/valu_bench6
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;

#define ADDS(T, G, F) asm volatile("v_add_f32 %0, %1, %2" : "=v"(T) : "s"(G), "v"(F))
#define MAX3(A, X, Y) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(A) : "v"(X), "v"(Y))
// t = (g.lo + f.lo, g.hi + f.lo) or, with HI, (g.lo + f.hi, g.hi + f.hi)
#define PKLO(T, G, F) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(T) : "s"(G), "v"(F))
#define PKHI(T, G, F) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(T) : "s"(G), "v"(F))

template <int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters, int seed, const int *__restrict__ gsrc) {
  int f0 = seed + threadIdx.x, f1 = f0 * 3, f2 = f0 * 5, f3 = f0 * 7;
  u64 fa = ((u64)(unsigned)f1 << 32) | (unsigned)f0, fb = ((u64)(unsigned)f3 << 32) | (unsigned)f2;
  int a0 = -(1 << 30), a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
  for (int it = 0; it < iters; ++it) {
    // 8 wave-uniform g values per iteration, in scalar registers
    const int *gp = gsrc + (it & 63) * 8;
    int g0 = gp[0], g1 = gp[1];
    int g2 = gp[2], g3 = gp[3];
    int g4 = gp[4], g5 = gp[5];
    int g6 = gp[6], g7 = gp[7];
    if (MODE == 0) {  // blocks of 2 g values x 4 registers; sums of a block consumed two blocks later
      int t[4][4][2];
#define SUMS(B, GA, GB) ADDS(t[B][0][0], GA, f0); ADDS(t[B][0][1], GB, f0); ADDS(t[B][1][0], GA, f1); ADDS(t[B][1][1], GB, f1); \
                        ADDS(t[B][2][0], GA, f2); ADDS(t[B][2][1], GB, f2); ADDS(t[B][3][0], GA, f3); ADDS(t[B][3][1], GB, f3)
#define MAXS(B, A0, A1, A2, A3) MAX3(A0, t[B][0][0], t[B][0][1]); MAX3(A1, t[B][1][0], t[B][1][1]); \
                                MAX3(A2, t[B][2][0], t[B][2][1]); MAX3(A3, t[B][3][0], t[B][3][1])
      SUMS(0, g0, g1); SUMS(1, g2, g3); MAXS(0, a0, a1, a2, a3); SUMS(2, g4, g5); MAXS(1, a4, a5, a6, a7);
      SUMS(3, g6, g7); MAXS(2, a0, a1, a2, a3); MAXS(3, a4, a5, a6, a7);
    } else if (MODE == 1) {
      int x, y;
#define TRI(F, GA, GB, A) ADDS(x, GA, F); ADDS(y, GB, F); MAX3(A, x, y)
      TRI(f0, g0, g1, a0); TRI(f1, g0, g1, a1); TRI(f2, g0, g1, a2); TRI(f3, g0, g1, a3);
      TRI(f0, g2, g3, a4); TRI(f1, g2, g3, a5); TRI(f2, g2, g3, a6); TRI(f3, g2, g3, a7);
      TRI(f0, g4, g5, a0); TRI(f1, g4, g5, a1); TRI(f2, g4, g5, a2); TRI(f3, g4, g5, a3);
      TRI(f0, g6, g7, a4); TRI(f1, g6, g7, a5); TRI(f2, g6, g7, a6); TRI(f3, g6, g7, a7);
    } else if (MODE == 2) {
      u64 ga = ((u64)(unsigned)g1 << 32) | (unsigned)g0, gb = ((u64)(unsigned)g3 << 32) | (unsigned)g2;
      u64 gc = ((u64)(unsigned)g5 << 32) | (unsigned)g4, gd = ((u64)(unsigned)g7 << 32) | (unsigned)g6;
      u64 t[4][4];
#define PSUMS(B, G) PKLO(t[B][0], G, fa); PKHI(t[B][1], G, fa); PKLO(t[B][2], G, fb); PKHI(t[B][3], G, fb)
#define PMAXS(B, A0, A1, A2, A3) MAX3(A0, (int)t[B][0], (int)(t[B][0] >> 32)); MAX3(A1, (int)t[B][1], (int)(t[B][1] >> 32)); \
                                 MAX3(A2, (int)t[B][2], (int)(t[B][2] >> 32)); MAX3(A3, (int)t[B][3], (int)(t[B][3] >> 32))
      PSUMS(0, ga); PSUMS(1, gb); PMAXS(0, a0, a1, a2, a3); PSUMS(2, gc); PMAXS(1, a4, a5, a6, a7);
      PSUMS(3, gd); PMAXS(2, a0, a1, a2, a3); PMAXS(3, a4, a5, a6, a7);
    } else if (MODE == 3) {  // 16 packed adds
      u64 ga = ((u64)(unsigned)g1 << 32) | (unsigned)g0, gb = ((u64)(unsigned)g3 << 32) | (unsigned)g2;
      u64 gc = ((u64)(unsigned)g5 << 32) | (unsigned)g4, gd = ((u64)(unsigned)g7 << 32) | (unsigned)g6;
      u64 t0, t1, t2, t3;
#define P4(G) PKLO(t0, G, fa); PKHI(t1, G, fa); PKLO(t2, G, fb); PKHI(t3, G, fb); \
              a0 ^= (int)t0; a1 ^= (int)(t1 >> 32); a2 ^= (int)t2; a3 ^= (int)(t3 >> 32)
      P4(ga); P4(gb); P4(gc); P4(gd);
    } else if (MODE == 4) {  // 32 scalar-operand adds (+ 16 xor to keep them alive)
      int x, y;
#define A2X(F, GA, GB, A) ADDS(x, GA, F); ADDS(y, GB, F); A ^= x ^ y
      A2X(f0, g0, g1, a0); A2X(f1, g0, g1, a1); A2X(f2, g0, g1, a2); A2X(f3, g0, g1, a3);
      A2X(f0, g2, g3, a4); A2X(f1, g2, g3, a5); A2X(f2, g2, g3, a6); A2X(f3, g2, g3, a7);
      A2X(f0, g4, g5, a0); A2X(f1, g4, g5, a1); A2X(f2, g4, g5, a2); A2X(f3, g4, g5, a3);
      A2X(f0, g6, g7, a4); A2X(f1, g6, g7, a5); A2X(f2, g6, g7, a6); A2X(f3, g6, g7, a7);
    } else {  // 16 max3
      MAX3(a0, f0, f1); MAX3(a1, f1, f2); MAX3(a2, f2, f3); MAX3(a3, f3, f0);
      MAX3(a4, f0, f1); MAX3(a5, f1, f2); MAX3(a6, f2, f3); MAX3(a7, f3, f0);
      MAX3(a0, f0, f2); MAX3(a1, f1, f3); MAX3(a2, f2, f0); MAX3(a3, f3, f1);
      MAX3(a4, f0, f2); MAX3(a5, f1, f3); MAX3(a6, f2, f0); MAX3(a7, f3, f1);
      a0 ^= g0 ^ g1 ^ g2 ^ g3 ^ g4 ^ g5 ^ g6 ^ g7;
    }
  }
  int s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  if (s == 0x12345678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
int run(const char *name, int *d_out, const int *d_g) {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  printf("%-58s", name);
  for (int wps : {2, 4, 6, 8}) {
    int blocks = cus * wps, iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 100, 1, d_g);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1, d_g);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("  w/SIMD=%d: %.2f", wps, 2.4e9 * (ms / 1e3) / ((double)iters * 16 * wps));
  }
  printf("\n");
  return 0;
}
int main() {
  int *d_out, *d_g;
  CHK(hipMalloc((void **)&d_out, 256 * 8 * 256 * sizeof(int)));
  CHK(hipMalloc((void **)&d_g, 64 * 8 * sizeof(int)));
  CHK(hipMemset(d_g, 0x3f, 64 * 8 * sizeof(int)));
  printf("cycles at 2.4 GHz per SIMD per 1/16 of an iteration (modes 0-2: per two candidates x 64 lanes; set-up of the 8 scalar g values included)\n");
  run<0>("0: 2 v_add_f32(s) + v_max3_f32, kernel order", d_out, d_g);
  run<1>("1: 2 v_add_f32(s) + v_max3_f32, max3 right after its adds", d_out, d_g);
  run<2>("2: v_pk_add_f32(s pair) + v_max3_f32, kernel order", d_out, d_g);
  run<3>("3: v_pk_add_f32 + xor (per pk_add)", d_out, d_g);
  run<4>("4: 2 v_add_f32(s) + 1 xor-pair (per 2 adds)", d_out, d_g);
  run<5>("5: v_max3_f32 (per max3)", d_out, d_g);
  return 0;
}
