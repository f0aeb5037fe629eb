// Micro-benchmark 10: issue ceilings of the candidate loop's instruction mix in CYCLES OF THE
// CLOCK THE CHIP ACTUALLY HELD, not of an assumed 2.4 GHz.  Every wave stamps s_memtime (shader
// cycles) and s_memrealtime (100 MHz) around its loop; the clock is the ratio, the median over
// waves.  Before the measured launch the same kernel runs for about 0.5 s (the clock ramp).
//
// Modes (per iteration the work of 16 q values x 4 registers of p = 64 candidate pairs per lane-column):
//   0  the kernel's long phase: 16 v_mov_b32 (scalar g -> vector), 64 v_add_f32, 32 v_max3_f32,
//      in the kernel's block order (A0 A1 M0 A2 M1 A3 M2 M3)
//   1  64 v_add_f32 alone (vector operands)
//   2  32 v_max3_f32 alone
//   3  64 v_add_f32 + 32 v_max3_f32 without the copies (vector g: the bare 2-add-1-max3 mix)
//   4  matrix-core form: one v_mfma_f32_32x32x16_bf16 (32 x 32 sums f[p] + g[q] as a K = 6 product
//      of bf16 pieces) + 8 v_max3_f32 on its 16 results per lane: 1024 candidates per MFMA
//   5  as 4 plus the six VALU instructions that split 32 values of g into bf16 pieces, once per 8 MFMAs
//   6  v_mfma_f32_32x32x16_bf16 alone, back to back      7, 8  as 6 and 4 with v_mfma_f32_32x32x8_bf16_1k
// Output: cycles per SIMD per 128 candidates (two candidates x 64 lanes), for 1, 2, 4, 8 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_bench10.hip -o tools/valu_bench10
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define ADDV(T, G, F) asm volatile("v_add_f32 %0, %1, %2" : "=v"(T) : "v"(G), "v"(F))
#define MOVS(T, G) asm volatile("v_mov_b32 %0, %1" : "=v"(T) : "s"(G))
#define MAX3(A, X, Y) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(A) : "v"(X), "v"(Y))

typedef __attribute__((ext_vector_type(8))) short frag_ab;   // 8 bf16
typedef __attribute__((ext_vector_type(16))) float frag_cd;  // 16 f32

struct Stamp {
  unsigned long long c0, c1, r0, r1;
};

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void k(int *out, int iters, int seed, const int *__restrict__ gsrc, Stamp *stamps) {
  int f[4], acc[8];
  for (int r = 0; r < 4; ++r) f[r] = __float_as_int((float)(seed + threadIdx.x * (2 * r + 1)));
  for (int i = 0; i < 8; ++i) acc[i] = __float_as_int(-1.0e9f);
  int vg[16];
  for (int i = 0; i < 16; ++i) vg[i] = __float_as_int((float)(gsrc[threadIdx.x & 7] + i));
  frag_ab a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (short)(0x3f80 + ((threadIdx.x + i) & 3));  // bf16 values near 1.0
    b[i] = (short)(0x3f80 + ((threadIdx.x * 3 + i) & 3));
  }
  typedef __attribute__((ext_vector_type(4))) short frag_ab4;  // 4 bf16: operand of the K = 8 form
  frag_ab4 a4, b4;
  for (int i = 0; i < 4; ++i) {
    a4[i] = a[i];
    b4[i] = b[i];
  }
  frag_cd d0 = {}, d1 = {};
  frag_cd mk;  // a tile that is not all zeros and stays in registers
  for (int i = 0; i < 16; ++i) mk[i] = (float)((threadIdx.x + i) & 3);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE <= 3) {
      const int *gp = gsrc + (it & 31) * 16;
      int t[4][4][4];
      auto sums = [&](int blk) {
        int g4[4];
        if (MODE == 0) {
          for (int i = 0; i < 4; ++i) MOVS(g4[i], gp[4 * blk + i]);
        } else {
          for (int i = 0; i < 4; ++i) g4[i] = vg[4 * blk + i];
        }
        if (MODE != 2) {
          for (int r = 0; r < 4; ++r) { ADDV(t[blk][r][0], g4[0], f[r]); ADDV(t[blk][r][1], g4[1], f[r]); }
          for (int r = 0; r < 4; ++r) { ADDV(t[blk][r][2], g4[2], f[r]); ADDV(t[blk][r][3], g4[3], f[r]); }
        } else {
          for (int r = 0; r < 4; ++r) for (int i = 0; i < 4; ++i) t[blk][r][i] = g4[i];
        }
      };
      auto maxima = [&](int blk) {
        if (MODE == 1) {
          // keep the sums alive without a max: fold them into f with one cheap op per block
          asm volatile("" ::"v"(t[blk][0][0]), "v"(t[blk][1][1]), "v"(t[blk][2][2]), "v"(t[blk][3][3]));
          return;
        }
        for (int r = 0; r < 4; ++r) MAX3(acc[r], t[blk][r][0], t[blk][r][1]);
        for (int r = 0; r < 4; ++r) MAX3(acc[4 + r], t[blk][r][2], t[blk][r][3]);
      };
      sums(0); sums(1); maxima(0); sums(2); maxima(1); sums(3); maxima(2); maxima(3);
      if (MODE != 0) for (int i = 0; i < 16; ++i) vg[i] ^= (it & 1);  // (kept out of the count below: 16 cheap ops)
    } else if (MODE == 6) {
      d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d1, 0, 0, 0);
    } else if (MODE == 7) {
      d0 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, d1, 0, 0, 0);
    } else if (MODE == 9) {
      // as 8, the matrix instruction on top of a tile held in registers (the mask of a diagonal tile)
      d0 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, mk, 0, 0, 0);
      for (int i = 0; i < 8; ++i) MAX3(acc[i], __float_as_int(d1[2 * i]), __float_as_int(d1[2 * i + 1]));
      d1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(b4, a4, mk, 0, 0, 0);
      for (int i = 0; i < 8; ++i) MAX3(acc[i], __float_as_int(d0[2 * i]), __float_as_int(d0[2 * i + 1]));
    } else if (MODE == 8) {
      const frag_cd z = {};
      d0 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, z, 0, 0, 0);
      for (int i = 0; i < 8; ++i) MAX3(acc[i], __float_as_int(d1[2 * i]), __float_as_int(d1[2 * i + 1]));
      d1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(b4, a4, z, 0, 0, 0);
      for (int i = 0; i < 8; ++i) MAX3(acc[i], __float_as_int(d0[2 * i]), __float_as_int(d0[2 * i + 1]));
    } else {
      // two MFMAs per iteration on two result tiles; the maxima of a tile are taken while the
      // other tile's MFMA executes
      const frag_cd z = {};
      d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0);
      for (int i = 0; i < 8; ++i) MAX3(acc[i], __float_as_int(d1[2 * i]), __float_as_int(d1[2 * i + 1]));
      d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, z, 0, 0, 0);
      for (int i = 0; i < 8; ++i) MAX3(acc[i], __float_as_int(d0[2 * i]), __float_as_int(d0[2 * i + 1]));
      if (MODE == 5 && (it & 3) == 0) {
        // split of 32 g values into three bf16 pieces: and, sub, and, sub, perm, perm
        int g = vg[0], hi, rem, mid, lo, p0, p1;
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hi) : "v"(g));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(rem) : "v"(g), "v"(hi));
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(mid) : "v"(rem));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(lo) : "v"(rem), "v"(mid));
        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(p0) : "v"(mid), "v"(hi), "s"(0x07060302));
        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(p1) : "v"(lo), "v"(hi), "s"(0x07060302));
        a[0] = (short)p0;
        a[1] = (short)(p0 >> 16);
        a[2] = (short)p1;
        vg[0] = g ^ 1;
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  int s = 0;
  for (int i = 0; i < 8; ++i) s ^= acc[i];
  for (int i = 0; i < 16; ++i) s ^= __float_as_int(d0[i]) ^ __float_as_int(d1[i]) ^ vg[i];
  if (s == 0x12345678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    Stamp st{c0, c1, r0, r1};
    stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = st;
  }
}

template <int MODE>
int run(const char *name, int *d_out, const int *d_g, Stamp *d_st, double cand_per_iter) {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  printf("%-62s", name);
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = cus * wps;
    // about 0.5 s of the same work first: an idle chip needs tens of milliseconds to reach the clock it sustains
    int iters = 4000;
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
      CHK(hipEventRecord(e0));
      hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1, d_g, d_st);
      CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1));
      CHK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 0) iters = (int)std::min(4.0e6, std::max(4000.0, iters * 500.0 / std::max(ms, 0.01f)));
    }
    const int miters = std::max(2000, iters / 25);  // about 20 ms, right behind the warm-up launch
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, miters, 1, d_g, d_st);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> st((size_t)blocks * 4);
    CHK(hipMemcpy(st.data(), d_st, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> mhz, cyc;
    for (const Stamp &s : st) {
      if (s.r1 > s.r0) mhz.push_back(100.0 * double(s.c1 - s.c0) / double(s.r1 - s.r0));
      cyc.push_back(double(s.c1 - s.c0));
    }
    std::sort(mhz.begin(), mhz.end());
    std::sort(cyc.begin(), cyc.end());
    const double clk = mhz.empty() ? 0.0 : mhz[mhz.size() / 2] * 1e6;
    // cycles of one SIMD per 128 candidates: wall time of the launch x measured clock / (work per SIMD)
    const double per128_wall = clk * (ms / 1e3) / ((double)miters * wps * cand_per_iter / 128.0);
    // the same from the waves' own cycle stamps (median wave): wave cycles / its iterations, times the waves sharing the SIMD
    const double per128_wave = cyc[cyc.size() / 2] / ((double)miters * cand_per_iter / 128.0) / wps;
    printf("  w=%d: %.2f (%.2f) @%.0f MHz", wps, per128_wall, per128_wave, clk / 1e6);
  }
  printf("\n");
  return 0;
}

int main() {
  int *d_out, *d_g;
  Stamp *d_st;
  CHK(hipMalloc((void **)&d_out, 256 * 8 * 256 * sizeof(int)));
  CHK(hipMalloc((void **)&d_g, 32 * 16 * sizeof(int)));
  CHK(hipMalloc((void **)&d_st, 256 * 8 * 4 * sizeof(Stamp)));
  {
    std::vector<float> g(32 * 16);
    for (size_t i = 0; i < g.size(); ++i) g[i] = (float)((i * 2654435761u) % 100000);
    CHK(hipMemcpy(d_g, g.data(), g.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  printf("cycles of one SIMD per 128 candidates (two candidates x 64 lanes) at the measured clock: from the launch's wall "
         "time, in brackets from the median wave's own cycle count / waves per SIMD; w = waves per SIMD\n");
  // candidates per iteration and wave: modes 0-3: 16 q x 4 registers x 64 lanes = 4096; modes 4-6: 2 MFMAs x 1024
  run<0>("0: kernel long phase: 16 v_mov + 64 v_add_f32 + 32 v_max3_f32", d_out, d_g, d_st, 4096);
  run<1>("1: 64 v_add_f32 alone", d_out, d_g, d_st, 4096);
  run<2>("2: 32 v_max3_f32 alone", d_out, d_g, d_st, 4096);
  run<3>("3: 64 v_add_f32 + 32 v_max3_f32 (+16 v_xor)", d_out, d_g, d_st, 4096);
  run<4>("4: v_mfma_f32_32x32x16_bf16 + 8 v_max3_f32 per 1024 candidates", d_out, d_g, d_st, 2048);
  run<5>("5: as 4 + 6 split instructions per 8 MFMAs", d_out, d_g, d_st, 2048);
  run<6>("6: v_mfma_f32_32x32x16_bf16 alone", d_out, d_g, d_st, 2048);
  run<7>("7: v_mfma_f32_32x32x8_bf16_1k alone (two-register operands)", d_out, d_g, d_st, 2048);
  run<8>("8: v_mfma_f32_32x32x8_bf16_1k + 8 v_max3_f32 per 1024 candidates", d_out, d_g, d_st, 2048);
  run<9>("9: as 8 with the C operand in registers (diagonal tiles)", d_out, d_g, d_st, 2048);
  return 0;
}
