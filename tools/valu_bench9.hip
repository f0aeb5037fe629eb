// Micro-benchmark 9: the long phase with g read from LDS (every lane the same address, one
// ds_read_b128 per 4 values) instead of scalar loads + v_mov_b32.  Per step of 16 g values:
//   0  s_load_dwordx16 + 16 v_mov_b32 + 64 v_add_f32 + 32 v_max3_f32   (the kernel's form)
//   1  4 ds_read_b128 (broadcast) + 64 v_add_f32 + 32 v_max3_f32, reads issued one step ahead
// Only the rate matters; results are meaningless.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_bench9.hip -o tools/valu_bench9
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define MOV32(T, G) asm volatile("v_mov_b32 %0, %1" : "=v"(T) : "s"(G))
#define ADDV(T, G, F) asm volatile("v_add_f32 %0, %1, %2" : "=v"(T) : "v"(G), "v"(F))
#define MAX3(A, X, Y) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(A) : "v"(X), "v"(Y))

template <int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters, int seed, const int *__restrict__ gsrc) {
  __shared__ int4 lg[256];  // 1024 g values
  for (int i = threadIdx.x; i < 256; i += 256) lg[i] = make_int4(gsrc[4 * i], gsrc[4 * i + 1], gsrc[4 * i + 2], gsrc[4 * i + 3]);
  __syncthreads();
  int f[4] = {seed + (int)threadIdx.x, 0, 0, 0};
  f[1] = f[0] * 3; f[2] = f[0] * 5; f[3] = f[0] * 7;
  int a[4][2];
  for (int r = 0; r < 4; ++r) a[r][0] = a[r][1] = -(1 << 30);
  int4 nx[4];
  if (MODE == 1)
    for (int b = 0; b < 4; ++b) nx[b] = lg[b];
  for (int it = 0; it < iters; ++it) {
    int v[4][4];
    if (MODE == 0) {
      const int *gp = gsrc + (it & 63) * 16;
      int g[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) g[i] = gp[i];
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) MOV32(v[b][i], g[4 * b + i]);
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) { v[b][0] = nx[b].x; v[b][1] = nx[b].y; v[b][2] = nx[b].z; v[b][3] = nx[b].w; }
      const int base = ((it + 1) & 63) * 4;  // wave-uniform address: a broadcast read
#pragma unroll
      for (int b = 0; b < 4; ++b) nx[b] = lg[base + b];
    }
    int t[4][4][4];
    auto sums = [&](int b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { ADDV(t[b][r][0], v[b][0], f[r]); ADDV(t[b][r][1], v[b][1], f[r]); }
#pragma unroll
      for (int r = 0; r < 4; ++r) { ADDV(t[b][r][2], v[b][2], f[r]); ADDV(t[b][r][3], v[b][3], f[r]); }
    };
    auto maxima = [&](int b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) MAX3(a[r][0], t[b][r][0], t[b][r][1]);
#pragma unroll
      for (int r = 0; r < 4; ++r) MAX3(a[r][1], t[b][r][2], t[b][r][3]);
    };
    sums(0); sums(1); maxima(0); sums(2); maxima(1); sums(3); maxima(2); maxima(3);
  }
  int s = 0;
  for (int r = 0; r < 4; ++r) s ^= a[r][0] ^ a[r][1];
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
  int *d_out, *d_g;
  CHK(hipMalloc((void **)&d_out, 256 * 8 * 256 * sizeof(int)));
  CHK(hipMalloc((void **)&d_g, 1024 * sizeof(int)));
  CHK(hipMemset(d_g, 0x3f, 1024 * sizeof(int)));
  printf("cycles at 2.4 GHz per SIMD per two candidates x 64 lanes (1/32 of a step of 16 g values, loop overhead included)\n");
  run<0>("0: s_load + v_mov_b32 + 2 v_add_f32 + v_max3_f32", d_out, d_g);
  run<1>("1: ds_read_b128 broadcast + 2 v_add_f32 + v_max3_f32", d_out, d_g);
  return 0;
}
