// Micro-benchmark of the resolve stage's dependent float32 chains: cycles per step of ONE wave per
// SIMD (the resolve kernels are latency-bound: a handful of waves on the whole chip), in cycles of
// the clock the chip really held (s_memtime / s_memrealtime), on an idle GPU and after 0.5 s of load.
//   0  registers only: c = max(v, c + y); v = v + x            (the chain of psa_first_p)
//   1  the same with the operands read from LDS as float2, lanes skewed, 8 reads in flight
//   2  registers only: c = c + y                               (the chain of psa_first_q)
//   3  registers only: the three-state step of psa_fwd with two wave-wide lane shifts
//   4  as 0, two independent chains interleaved in one wave (does the second one ride for free?)
//   5  registers only: the step of psa_fwd as shipped (one packed add, two maxima with a lane-shifted
//      operand, one add)
//   6  issue cost of 8 independent v_add_f32 with a wave_shl:1 operand   7  of 8 independent v_pk_add_f32
//   8  of 8 independent v_add_f32                                        9  of 8 independent v_max_f32, row_shl:1 operand
// and a bit-for-bit comparison of v_pk_add_f32 with v_add_f32 on 2^22 operand pairs, subnormals included.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/chain_bench.hip -o tools/chain_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include <unistd.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Stamp {
  unsigned long long c0, c1, r0, r1;
};

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float lane_shl1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_shl1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_shr1(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xf, 0xf, true));
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int nsteps, const float *src, Stamp *stamps) {
  __shared__ float2 lds[4096 + 512];
  for (int i = threadIdx.x; i < 4096 + 512; i += 256) lds[i] = make_float2(src[i & 1023], src[(i * 7) & 1023]);
  __syncthreads();
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x;
  float v = src[tid], c = -1e30f, x = src[tid + 256], y = src[tid + 512];
  float v2 = src[tid + 1], c2 = -1e30f;
  float a = 0.f, b = -1e30f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (MODE == 0) {
    for (int t = 0; t < nsteps; t += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c = fmaxf(v, c + y);
        v = v + x;
        asm volatile("" : "+v"(x), "+v"(y));
      }
    }
  } else if (MODE == 1) {
    const float2 *dp = lds + tid + 1;
    float2 yy[8], zz[8];
    int t = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) yy[u] = dp[u];
    for (; t + 24 <= nsteps; t += 16) {
#pragma unroll
      for (int u = 0; u < 8; ++u) zz[u] = dp[t + 8 + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c = fmaxf(v, c + yy[u].y);
        v = v + yy[u].x;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) yy[u] = dp[t + 16 + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c = fmaxf(v, c + zz[u].y);
        v = v + zz[u].x;
      }
    }
  } else if (MODE == 2) {
    for (int t = 0; t < nsteps; t += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c = c + y;
        asm volatile("" : "+v"(y));
      }
    }
  } else if (MODE == 3) {
    for (int t = 0; t < nsteps; t += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float x1 = lane_shr1(x), x2 = lane_shr1(x1);
        const float nc = fmaxf(b, c + x2);
        const float nb = fmaxf(a, b + x1);
        a = a + x;
        b = nb;
        c = nc;
        asm volatile("" : "+v"(x));
      }
    }
    v = a + b;
  } else if (MODE == 5) {
    f2 bc = {b, c};
    for (int t = 0; t < nsteps; t += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f2 tt = bc + (f2){x, x};
        const float nc = __builtin_fmaxf(lane_shl1(bc.x), tt.y);
        const float nb = __builtin_fmaxf(lane_shl1(a), tt.x);
        a = a + x;
        bc = (f2){nb, nc};
        asm volatile("" : "+v"(x));
      }
    }
    v = a + bc.x;
    c = bc.y;
  } else if (MODE >= 6 && MODE <= 9) {
    float r[8];
    f2 q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      r[u] = src[tid + u];
      q[u] = (f2){r[u], y};
    }
    for (int t = 0; t < nsteps; t += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (MODE == 6) r[u] = lane_shl1(r[u]) + x;
        if (MODE == 7) q[u] = q[u] + (f2){x, y};
        if (MODE == 8) r[u] = r[u] + x;
        if (MODE == 9) r[u] = __builtin_fmaxf(row_shl1(r[u]), x);
      }
      asm volatile("" : "+v"(x), "+v"(y));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) v += r[u] + q[u].x + q[u].y;
  } else {
    for (int t = 0; t < nsteps; t += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c = fmaxf(v, c + y);
        c2 = fmaxf(v2, c2 + x);
        v = v + x;
        v2 = v2 + y;
        asm volatile("" : "+v"(x), "+v"(y));
      }
    }
    v += v2 + c2;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + tid] = v + c;
  if ((tid & 63) == 0) {
    Stamp s{c0, c1, r0, r1};
    stamps[blockIdx.x * 4 + (tid >> 6)] = s;
  }
}

// v_pk_add_f32 against v_add_f32 on arbitrary bit patterns (NaN results compared as NaN)
__global__ void pk_check(unsigned *mismatch, unsigned seed) {
  unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed;
  unsigned bad = 0;
  for (int i = 0; i < 64; ++i) {
    h = h * 1664525u + 1013904223u;
    unsigned ua = h;
    h = h * 1664525u + 1013904223u;
    unsigned ub = h;
    if (i & 1) {  // small exponents: subnormal operands and results
      ua &= 0x80ffffffu;
      ub = (ub & 0x80ffffffu) | ((i & 2) ? 0x00800000u : 0u);
    }
    if (i % 8 == 4) ub = ua ^ 0x80000001u;  // near cancellation
    const float fa = __uint_as_float(ua), fb = __uint_as_float(ub);
    float s1;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(fa), "v"(fb));
    f2 pa = {fa, fb}, pb = {fb, fa}, ps;
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(ps) : "v"(pa), "v"(pb));
    const unsigned u1 = __float_as_uint(s1), u2 = __float_as_uint(ps.x), u3 = __float_as_uint(ps.y);
    const bool nan1 = (u1 & 0x7fffffffu) > 0x7f800000u;
    if (nan1 ? ((u2 & 0x7fffffffu) <= 0x7f800000u) : (u1 != u2 || u1 != u3)) ++bad;
  }
  if (bad) atomicAdd(mismatch, bad);
}

__global__ void burn(float *out, int iters) {
  float a = threadIdx.x, b = 1.0001f;
  for (int i = 0; i < iters; ++i) a = a * b + 0.5f;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

template <int MODE>
int run(const char *name, float *d_out, const float *d_src, Stamp *d_st, float *d_burn) {
  const int nsteps = 4096;
  printf("%-64s", name);
  for (int warm = 0; warm < 2; ++warm) {
    for (int grid : {1, 8}) {
      if (warm) {
        hipLaunchKernelGGL(burn, dim3(4096), dim3(256), 0, 0, d_burn, 600000);  // about half a second of load
        CHK(hipGetLastError());
      } else {
        CHK(hipDeviceSynchronize());
        usleep(200000);
      }
      hipEvent_t e0, e1;
      CHK(hipEventCreate(&e0));
      CHK(hipEventCreate(&e1));
      CHK(hipEventRecord(e0));
      hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, nsteps, d_src, d_st);
      CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1));
      float ms = 0;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<Stamp> st(grid * 4);
      CHK(hipMemcpy(st.data(), d_st, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
      std::vector<double> cyc, mhz;
      for (auto &s : st) {
        cyc.push_back(double(s.c1 - s.c0) / nsteps);
        mhz.push_back(double(s.c1 - s.c0) / std::max<double>(1, double(s.r1 - s.r0)) * 100.0);
      }
      std::sort(cyc.begin(), cyc.end());
      std::sort(mhz.begin(), mhz.end());
      printf("  %s grid=%d: %.1f cyc/step @%.0f MHz (%.1f us)", warm ? "warm" : "idle", grid, cyc[cyc.size() / 2],
             mhz[mhz.size() / 2], ms * 1e3);
    }
  }
  printf("\n");
  return 0;
}

int main() {
  float *d_out, *d_src, *d_burn;
  Stamp *d_st;
  CHK(hipMalloc(&d_out, 8 * 256 * sizeof(float)));
  CHK(hipMalloc(&d_src, 1024 * sizeof(float)));
  CHK(hipMalloc(&d_burn, 4096 * 256 * sizeof(float)));
  CHK(hipMalloc(&d_st, 32 * sizeof(Stamp)));
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = float((i * 37) % 11) - 4.0f;
  CHK(hipMemcpy(d_src, h.data(), 1024 * sizeof(float), hipMemcpyHostToDevice));
  if (run<0>("0: c = max(v, c + y); v += x (registers)", d_out, d_src, d_st, d_burn)) return 1;
  if (run<1>("1: the same, operands from LDS (float2, skewed, 8 in flight)", d_out, d_src, d_st, d_burn)) return 1;
  if (run<2>("2: c += y (registers)", d_out, d_src, d_st, d_burn)) return 1;
  if (run<3>("3: three-state step with two lane shifts (registers)", d_out, d_src, d_st, d_burn)) return 1;
  if (run<4>("4: two independent chains of kind 0 in one wave", d_out, d_src, d_st, d_burn)) return 1;
  if (run<5>("5: psa_fwd's step: pk_add, 2 max with shifted operand, add", d_out, d_src, d_st, d_burn)) return 1;
  if (run<6>("6: 8 independent v_add_f32 wave_shl:1 per 8 'steps'", d_out, d_src, d_st, d_burn)) return 1;
  if (run<7>("7: 8 independent v_pk_add_f32", d_out, d_src, d_st, d_burn)) return 1;
  if (run<8>("8: 8 independent v_add_f32", d_out, d_src, d_st, d_burn)) return 1;
  if (run<9>("9: 8 independent v_max_f32 row_shl:1", d_out, d_src, d_st, d_burn)) return 1;
  unsigned *d_bad, h_bad = 0;
  CHK(hipMalloc(&d_bad, 4));
  CHK(hipMemset(d_bad, 0, 4));
  hipLaunchKernelGGL(pk_check, dim3(256), dim3(256), 0, 0, d_bad, 12345u);
  CHK(hipMemcpy(&h_bad, d_bad, 4, hipMemcpyDeviceToHost));
  printf("v_pk_add_f32 against v_add_f32 on %d operand pairs (subnormals, cancellation, infinities, NaN): %u mismatches\n",
         256 * 256 * 64, h_bad);
  return 0;
}
