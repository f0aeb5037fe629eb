// Micro-benchmark 7: where the g operand of the candidate loop's adds comes from.
//   0  v_add_f32 with scalar g, g alternating between consecutive adds (kernel order)
//   1  v_add_f32 with scalar g, the four adds of one g value consecutive
//   2  one v_mov_b32 per g value into a vector register, then four vector-operand adds
//   3  v_add_f32 with g in vector registers (reference rate; not available to the kernel for free)
//   4  as 0 with v_add_u32 / v_max3_i32
//   5  as 3 with v_add_u32 / v_max3_i32
// Each mode: 32 adds + 16 max3 per iteration (16 x two candidates), 8 scalar g values per iteration.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_bench7.hip -o tools/valu_bench7
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define ADDS(T, G, F) asm volatile("v_add_f32 %0, %1, %2" : "=v"(T) : "s"(G), "v"(F))
#define ADDV(T, G, F) asm volatile("v_add_f32 %0, %1, %2" : "=v"(T) : "v"(G), "v"(F))
#define MOVS(T, G) asm volatile("v_mov_b32 %0, %1" : "=v"(T) : "s"(G))
#define MAX3(A, X, Y) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(A) : "v"(X), "v"(Y))
#define IADDS(T, G, F) asm volatile("v_add_u32 %0, %1, %2" : "=v"(T) : "s"(G), "v"(F))
#define IADDV(T, G, F) asm volatile("v_add_u32 %0, %1, %2" : "=v"(T) : "v"(G), "v"(F))
#define IMAX3(A, X, Y) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(A) : "v"(X), "v"(Y))

template <int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters, int seed, const int *__restrict__ gsrc) {
  int f0 = seed + threadIdx.x, f1 = f0 * 3, f2 = f0 * 5, f3 = f0 * 7;
  int a0 = -(1 << 30), a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
  int v0 = gsrc[threadIdx.x & 7], v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  for (int it = 0; it < iters; ++it) {
    const int *gp = gsrc + (it & 63) * 8;
    int g0 = gp[0], g1 = gp[1], g2 = gp[2], g3 = gp[3], g4 = gp[4], g5 = gp[5], g6 = gp[6], g7 = gp[7];
    int t[4][4][2];
    // block B: g values GA, GB against f0..f3; sums of a block are consumed two blocks later
#define RUN(ADD, MX, SUMS)                                                                             \
    SUMS(ADD, 0, g0, g1); SUMS(ADD, 1, g2, g3); MAXS(MX, 0, a0, a1, a2, a3); SUMS(ADD, 2, g4, g5);       \
    MAXS(MX, 1, a4, a5, a6, a7); SUMS(ADD, 3, g6, g7); MAXS(MX, 2, a0, a1, a2, a3); MAXS(MX, 3, a4, a5, a6, a7)
#define SUMS_ALT(ADD, B, GA, GB) ADD(t[B][0][0], GA, f0); ADD(t[B][0][1], GB, f0); ADD(t[B][1][0], GA, f1); ADD(t[B][1][1], GB, f1); \
                                 ADD(t[B][2][0], GA, f2); ADD(t[B][2][1], GB, f2); ADD(t[B][3][0], GA, f3); ADD(t[B][3][1], GB, f3)
#define SUMS_SAME(ADD, B, GA, GB) ADD(t[B][0][0], GA, f0); ADD(t[B][1][0], GA, f1); ADD(t[B][2][0], GA, f2); ADD(t[B][3][0], GA, f3); \
                                  ADD(t[B][0][1], GB, f0); ADD(t[B][1][1], GB, f1); ADD(t[B][2][1], GB, f2); ADD(t[B][3][1], GB, f3)
#define SUMS_MOV(ADD, B, GA, GB) { int ma, mb; MOVS(ma, GA); MOVS(mb, GB);                                     \
                                  ADD(t[B][0][0], ma, f0); ADD(t[B][0][1], mb, f0); ADD(t[B][1][0], ma, f1); ADD(t[B][1][1], mb, f1); \
                                  ADD(t[B][2][0], ma, f2); ADD(t[B][2][1], mb, f2); ADD(t[B][3][0], ma, f3); ADD(t[B][3][1], mb, f3); }
#define MAXS(MX, B, A0, A1, A2, A3) MX(A0, t[B][0][0], t[B][0][1]); MX(A1, t[B][1][0], t[B][1][1]); \
                                    MX(A2, t[B][2][0], t[B][2][1]); MX(A3, t[B][3][0], t[B][3][1])
    if (MODE == 0) { RUN(ADDS, MAX3, SUMS_ALT); }
    if (MODE == 1) { RUN(ADDS, MAX3, SUMS_SAME); }
    if (MODE == 2) { RUN(ADDV, MAX3, SUMS_MOV); }
    if (MODE == 3 || MODE == 5) {
      // the scalar values still steer the vector ones so that the loads stay in the loop
      v0 ^= g0; v1 ^= g1; v2 ^= g2; v3 ^= g3; v4 ^= g4; v5 ^= g5; v6 ^= g6; v7 ^= g7;
#define g0 v0
#define g1 v1
#define g2 v2
#define g3 v3
#define g4 v4
#define g5 v5
#define g6 v6
#define g7 v7
      if (MODE == 3) { RUN(ADDV, MAX3, SUMS_ALT); } else { RUN(IADDV, IMAX3, SUMS_ALT); }
#undef g0
#undef g1
#undef g2
#undef g3
#undef g4
#undef g5
#undef g6
#undef g7
    }
    if (MODE == 4) { RUN(IADDS, IMAX3, SUMS_ALT); }
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
  printf("cycles at 2.4 GHz per SIMD per two candidates x 64 lanes (1/16 of an iteration, loop overhead included)\n");
  run<0>("0: f32, scalar g, alternating", d_out, d_g);
  run<1>("1: f32, scalar g, four adds per g consecutive", d_out, d_g);
  run<2>("2: f32, v_mov per g + vector adds", d_out, d_g);
  run<3>("3: f32, vector g (8 extra v_xor per iteration)", d_out, d_g);
  run<4>("4: i32, scalar g, alternating", d_out, d_g);
  run<5>("5: i32, vector g (8 extra v_xor per iteration)", d_out, d_g);
  return 0;
}
