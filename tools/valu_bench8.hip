// Micro-benchmark 8: the long phase of the candidate loop (4 registers f, 16 values of g per step
// from one scalar load) in three instruction forms, per step of 16 g values = 32 x two candidates:
//   0  16 v_mov_b32 + 64 v_add_f32 (vector operands) + 32 v_max3_f32     (the kernel's form)
//   1  8 v_mov_b64 + 32 v_pk_add_f32 (vector operands, f broadcast by op_sel) + 32 v_max3_f32
//   2  32 v_pk_add_f32 with the scalar pair (g[i], g[i+1]) as operand + 32 v_max3_f32
// Only the rate matters; results are meaningless.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_bench8.hip -o tools/valu_bench8
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;

#define MOV32(T, G) asm volatile("v_mov_b32 %0, %1" : "=v"(T) : "s"(G))
#define MOV64(T, G) asm volatile("v_mov_b64 %0, %1" : "=v"(T) : "s"(G))
#define ADDV(T, G, F) asm volatile("v_add_f32 %0, %1, %2" : "=v"(T) : "v"(G), "v"(F))
#define MAX3(A, X, Y) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(A) : "v"(X), "v"(Y))
// t = (g.lo + f.lo, g.hi + f.lo) or, with HI, (g.lo + f.hi, g.hi + f.hi)
#define PKLO_V(T, G, F) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(T) : "v"(G), "v"(F))
#define PKHI_V(T, G, F) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(T) : "v"(G), "v"(F))
#define PKLO_S(T, G, F) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(T) : "s"(G), "v"(F))
#define PKHI_S(T, G, F) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(T) : "s"(G), "v"(F))
#define LO(X) ((int)(X))
#define HI(X) ((int)((X) >> 32))

template <int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters, int seed, const u64 *__restrict__ gsrc) {
  int f0 = seed + threadIdx.x, f1 = f0 * 3, f2 = f0 * 5, f3 = f0 * 7;
  u64 fa = ((u64)(unsigned)f1 << 32) | (unsigned)f0, fb = ((u64)(unsigned)f3 << 32) | (unsigned)f2;
  int a[4][2];
  for (int r = 0; r < 4; ++r) a[r][0] = a[r][1] = -(1 << 30);
  for (int it = 0; it < iters; ++it) {
    const u64 *gp = gsrc + (it & 63) * 8;  // 16 values = 8 pairs, wave-uniform: one s_load_dwordx16
    u64 g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = gp[i];
    if (MODE == 0) {
      int t[4][4][4];
      auto sums = [&](int b) {  // block b: g values 4b .. 4b+3
        int v[4];
        MOV32(v[0], LO(g[2 * b])); MOV32(v[1], HI(g[2 * b])); MOV32(v[2], LO(g[2 * b + 1])); MOV32(v[3], HI(g[2 * b + 1]));
        const int f[4] = {f0, f1, f2, f3};
#pragma unroll
        for (int r = 0; r < 4; ++r) { ADDV(t[b][r][0], v[0], f[r]); ADDV(t[b][r][1], v[1], f[r]); }
#pragma unroll
        for (int r = 0; r < 4; ++r) { ADDV(t[b][r][2], v[2], f[r]); ADDV(t[b][r][3], v[3], f[r]); }
      };
      auto maxima = [&](int b) {
#pragma unroll
        for (int r = 0; r < 4; ++r) MAX3(a[r][0], t[b][r][0], t[b][r][1]);
#pragma unroll
        for (int r = 0; r < 4; ++r) MAX3(a[r][1], t[b][r][2], t[b][r][3]);
      };
      sums(0); sums(1); maxima(0); sums(2); maxima(1); sums(3); maxima(2); maxima(3);
    } else {
      u64 t[4][4][2];
      auto sums = [&](int b) {
        if (MODE == 1) {
          u64 v0, v1;
          MOV64(v0, g[2 * b]); MOV64(v1, g[2 * b + 1]);
          PKLO_V(t[b][0][0], v0, fa); PKHI_V(t[b][1][0], v0, fa); PKLO_V(t[b][2][0], v0, fb); PKHI_V(t[b][3][0], v0, fb);
          PKLO_V(t[b][0][1], v1, fa); PKHI_V(t[b][1][1], v1, fa); PKLO_V(t[b][2][1], v1, fb); PKHI_V(t[b][3][1], v1, fb);
        } else {
          PKLO_S(t[b][0][0], g[2 * b], fa); PKHI_S(t[b][1][0], g[2 * b], fa); PKLO_S(t[b][2][0], g[2 * b], fb); PKHI_S(t[b][3][0], g[2 * b], fb);
          PKLO_S(t[b][0][1], g[2 * b + 1], fa); PKHI_S(t[b][1][1], g[2 * b + 1], fa); PKLO_S(t[b][2][1], g[2 * b + 1], fb); PKHI_S(t[b][3][1], g[2 * b + 1], fb);
        }
      };
      auto maxima = [&](int b) {
#pragma unroll
        for (int r = 0; r < 4; ++r) MAX3(a[r][0], LO(t[b][r][0]), HI(t[b][r][0]));
#pragma unroll
        for (int r = 0; r < 4; ++r) MAX3(a[r][1], LO(t[b][r][1]), HI(t[b][r][1]));
      };
      sums(0); sums(1); maxima(0); sums(2); maxima(1); sums(3); maxima(2); maxima(3);
    }
  }
  int s = 0;
  for (int r = 0; r < 4; ++r) s ^= a[r][0] ^ a[r][1];
  if (s == 0x12345678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
int run(const char *name, int *d_out, const u64 *d_g) {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  printf("%-66s", name);
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
    printf("  w/SIMD=%d: %.2f", wps, 2.4e9 * (ms / 1e3) / ((double)iters * 32 * wps));
  }
  printf("\n");
  return 0;
}
int main() {
  int *d_out;
  u64 *d_g;
  CHK(hipMalloc((void **)&d_out, 256 * 8 * 256 * sizeof(int)));
  CHK(hipMalloc((void **)&d_g, 64 * 8 * sizeof(u64)));
  CHK(hipMemset(d_g, 0x3f, 64 * 8 * sizeof(u64)));
  printf("cycles at 2.4 GHz per SIMD per two candidates x 64 lanes (1/32 of a step of 16 g values, loop overhead included)\n");
  run<0>("0: v_mov_b32 + 2 v_add_f32 (vector) + v_max3_f32", d_out, d_g);
  run<1>("1: v_mov_b64 + v_pk_add_f32 (vector) + v_max3_f32", d_out, d_g);
  run<2>("2: v_pk_add_f32 (scalar pair) + v_max3_f32", d_out, d_g);
  return 0;
}
