// Prototype of a wave-granular enumeration kernel: f in vector registers, g read through
// scalar loads, no LDS and no barriers.  Rows of precomputed f/g values stand in for the
// pre-pass.  Build: hipcc -O3 --offload-arch=gfx950 -o wave_proto wave_proto.hip
// Run:   ./wave_proto [rows] [len2] [R] [reps]
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      std::exit(1);                                                                    \
    }                                                                                  \
  } while (0)

__device__ long long *g_trace;
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

template <int R, int K>
__device__ __forceinline__ void score16(const int (&g)[16], const int (&f)[R], int (&acc)[R][2]) {
#pragma unroll
  for (int i = 0; i < 16; i += 4) {
#pragma unroll
    for (int r = 0; r < K; ++r) acc[r][0] = max3i(acc[r][0], f[r] + g[i], f[r] + g[i + 1]);
#pragma unroll
    for (int r = 0; r < K; ++r) acc[r][1] = max3i(acc[r][1], f[r] + g[i + 2], f[r] + g[i + 3]);
  }
}

typedef int int16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int vadd(int g, int f) {
  int t;
  asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "s"(g), "v"(f));
  return t;
}
__device__ __forceinline__ void vmax3(int &a, int x, int y) { asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y)); }

template <int R, int K>
__device__ __forceinline__ void full(const int *__restrict__ G, int qa, int qb, const int (&f)[R], int (&acc)[R][2]) {
  if (qa >= qb) return;
#if FULLV == 0
  int nx[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) nx[i] = G[qa + i];
  for (int q = qa; q < qb; q += 16) {
    int g[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) g[i] = nx[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) nx[i] = G[q + 16 + i];  // the last read lands in the slack after the row
    score16<R, K>(g, f, acc);
  }
#elif FULLV == 1
  for (int q = qa; q < qb; q += 16) {
    int g[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) g[i] = G[q + i];
    score16<R, K>(g, f, acc);
  }
#elif FULLV == 3
  for (int q = qa; q < qb; q += 16) {
    int g[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) g[i] = G[q + i];
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
      int t[K][4];
#pragma unroll
      for (int r = 0; r < K; ++r) {
        t[r][0] = vadd(g[i], f[r]);
        t[r][1] = vadd(g[i + 1], f[r]);
      }
#pragma unroll
      for (int r = 0; r < K; ++r) {
        t[r][2] = vadd(g[i + 2], f[r]);
        t[r][3] = vadd(g[i + 3], f[r]);
      }
#pragma unroll
      for (int r = 0; r < K; ++r) vmax3(acc[r][0], t[r][0], t[r][1]);
#pragma unroll
      for (int r = 0; r < K; ++r) vmax3(acc[r][1], t[r][2], t[r][3]);
    }
  }
#elif FULLV == 5
  // blocks of 4 g values: sums of block b are consumed two blocks later
  for (int q = qa; q < qb; q += 16) {
    int g[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) g[i] = G[q + i];
    int t[4][K][4];
    auto A = [&](int b) {
#pragma unroll
      for (int r = 0; r < K; ++r) {
        t[b][r][0] = vadd(g[4 * b], f[r]);
        t[b][r][1] = vadd(g[4 * b + 1], f[r]);
      }
#pragma unroll
      for (int r = 0; r < K; ++r) {
        t[b][r][2] = vadd(g[4 * b + 2], f[r]);
        t[b][r][3] = vadd(g[4 * b + 3], f[r]);
      }
    };
    auto M = [&](int b) {
#pragma unroll
      for (int r = 0; r < K; ++r) vmax3(acc[r][0], t[b][r][0], t[b][r][1]);
#pragma unroll
      for (int r = 0; r < K; ++r) vmax3(acc[r][1], t[b][r][2], t[b][r][3]);
    };
    A(0); A(1); M(0); A(2); M(1); A(3); M(2); M(3);
  }
#elif FULLV == 6
  // as 5, 32 g values per step (two scalar loads in flight)
  for (int q = qa; q < qb; q += 32) {
    int g[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) g[i] = G[q + i];
    int t[8][K][4];
    auto A = [&](int b) {
#pragma unroll
      for (int r = 0; r < K; ++r) {
        t[b][r][0] = vadd(g[4 * b], f[r]);
        t[b][r][1] = vadd(g[4 * b + 1], f[r]);
      }
#pragma unroll
      for (int r = 0; r < K; ++r) {
        t[b][r][2] = vadd(g[4 * b + 2], f[r]);
        t[b][r][3] = vadd(g[4 * b + 3], f[r]);
      }
    };
    auto M = [&](int b) {
#pragma unroll
      for (int r = 0; r < K; ++r) vmax3(acc[r][0], t[b][r][0], t[b][r][1]);
#pragma unroll
      for (int r = 0; r < K; ++r) vmax3(acc[r][1], t[b][r][2], t[b][r][3]);
    };
    A(0); A(1); M(0); A(2); M(1); A(3); M(2); A(4); M(3); A(5); M(4); A(6); M(5); A(7); M(6); M(7);
  }
#elif FULLV == 4
  // add, add, max3 interleaved; a max3 consumes the sums of the previous pair of g values
  for (int q = qa; q < qb; q += 16) {
    int g[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) g[i] = G[q + i];
    int t[K][2], n[K][2];
#pragma unroll
    for (int r = 0; r < K; ++r) {
      t[r][0] = vadd(g[0], f[r]);
      t[r][1] = vadd(g[1], f[r]);
    }
#pragma unroll
    for (int i = 2; i < 16; i += 2) {
#pragma unroll
      for (int r = 0; r < K; ++r) {
        n[r][0] = vadd(g[i], f[r]);
        n[r][1] = vadd(g[i + 1], f[r]);
        vmax3(acc[r][(i >> 1) & 1], t[r][0], t[r][1]);
      }
#pragma unroll
      for (int r = 0; r < K; ++r) {
        t[r][0] = n[r][0];
        t[r][1] = n[r][1];
      }
    }
#pragma unroll
    for (int r = 0; r < K; ++r) vmax3(acc[r][0], t[r][0], t[r][1]);
  }
#endif
}

template <int R, int K>
struct Phases {
  __device__ static __forceinline__ void run(const int *__restrict__ G, const int (&pc)[R], int qend, const int (&f)[R],
                                             int (&acc)[R][2]) {
    Phases<R, K - 1>::run(G, pc, qend, f, acc);
    int qa = 64 * (pc[K - 1] + 1);
    int qb = qend;
    if (K < R) qb = min(64 * (pc[K < R ? K : 0] + 1), qend);
    full<R, K>(G, qa, qb, f, acc);
  }
};
template <int R>
struct Phases<R, 0> {
  __device__ static __forceinline__ void run(const int *__restrict__, const int (&)[R], int, const int (&)[R],
                                             int (&)[R][2]) {}
};

template <int R>
__device__ __forceinline__ void unit_body(const int *__restrict__ FG, int nrows, int nruns, int NW, int len2,
                                                   unsigned long long *__restrict__ unit_best, int wmajor, int u) {
  const int lane = threadIdx.x & 63;
  if (u >= nrows * NW) return;
  const long long t_start = wall_clock64();
  int row, w;
  if (wmajor) {
    w = u / nrows;
    row = u - w * nrows;
  } else {
    row = u / NW;
    w = u - row * NW;
  }
  const int stride = 128 * nruns + 64;
  const int *__restrict__ F = FG + (size_t)row * stride;
  const int *__restrict__ G = F + 64 * nruns;
  const int qend = 64 * nruns;
  int pc[R], f[R], acc[R][2];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    pc[r] = w * R + r;
    f[r] = pc[r] < nruns ? F[64 * pc[r] + lane] : 0;
    acc[r][0] = acc[r][1] = INT_MIN;
  }
  Phases<R, R>::run(G, pc, qend, f, acc);
  // diagonal runs: q = 64*pc[r] + i pairs with lanes < i
#pragma unroll 1
  for (int i0 = 0; i0 < 64; i0 += 8) {
    int g[R][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int base = pc[r] < nruns ? 64 * pc[r] + i0 : 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) g[r][i] = G[base + i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (lane < i0 + i) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r][i & 1] = max(acc[r][i & 1], f[r] + g[r][i]);
      }
    }
  }
  int bestv = INT_MIN;
  uint32_t bestp = 0x7fffffffu;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int p = 64 * pc[r] + lane;
    const int a = max(acc[r][0], acc[r][1]);
    if (p <= len2 - 2 && a > bestv) {
      bestv = a;
      bestp = (uint32_t)p;
    }
  }
  unsigned long long key = ((unsigned long long)((uint32_t)bestv ^ 0x80000000u) << 32) | (0xffffffffu - bestp);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    uint32_t lo = __shfl_xor((uint32_t)key, m), hi = __shfl_xor((uint32_t)(key >> 32), m);
    unsigned long long o = ((unsigned long long)hi << 32) | lo;
    key = o > key ? o : key;
  }
  if (lane == 0) unit_best[row * NW + w] = key;
  if (lane == 0 && g_trace) {
    g_trace[3 * u] = t_start;
    g_trace[3 * u + 1] = wall_clock64();
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    g_trace[3 * u + 2] = hw;
  }
}


template <int R>
__global__ __launch_bounds__(64, 8) void wave_enum(const int *__restrict__ FG, int nrows, int nruns, int NW, int len2,
                                                   unsigned long long *__restrict__ unit_best, int wmajor, unsigned *) {
  unit_body<R>(FG, nrows, nruns, NW, len2, unit_best, wmajor, uni(blockIdx.x));
}
template <int R>
__global__ __launch_bounds__(512, 8) void wave_enum_b(const int *__restrict__ FG, int nrows, int nruns, int NW, int len2,
                                                      unsigned long long *__restrict__ unit_best, int wmajor, unsigned *) {
  unit_body<R>(FG, nrows, nruns, NW, len2, unit_best, wmajor, uni(blockIdx.x * 8 + (threadIdx.x >> 6)));
}
template <int R>
__global__ __launch_bounds__(512, 8) void wave_enum_c(const int *__restrict__ FG, int nrows, int nruns, int NW, int len2,
                                                      unsigned long long *__restrict__ unit_best, int wmajor,
                                                      unsigned *counters) {
  const int lane = threadIdx.x & 63;
  const int grp = blockIdx.x % NC;
  unsigned *counter = counters + grp * 64;
  unsigned t = 0;
  if (lane == 0) t = atomicAdd(counter, 1u);
  int u = uni((int)t) * NC + grp;
  const int nunits = nrows * NW;
  while (u < nunits) {
    if (lane == 0) t = atomicAdd(counter, 1u);
    unit_body<R>(FG, nrows, nruns, NW, len2, unit_best, wmajor, u);
    u = uni((int)t) * NC + grp;
  }
}
template <int R>
static float run(const int *dFG, int rows, int nruns, int len2, unsigned long long *dbest, int reps, int *NWout, int wmajor, int mode, int percu) {
  const int NW = (nruns + R - 1) / R;
  *NWout = NW;
  const int nunits = rows * NW;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  unsigned *dc;
  CK(hipMalloc((void **)&dc, NC * 64 * 4));
  auto launch = [&]() {
    if (mode == 0) hipLaunchKernelGGL(wave_enum<R>, dim3(nunits), dim3(64), 0, 0, dFG, rows, nruns, NW, len2, dbest, wmajor, dc);
    if (mode == 1) hipLaunchKernelGGL(wave_enum_b<R>, dim3((nunits + 7) / 8), dim3(512), 0, 0, dFG, rows, nruns, NW, len2, dbest, wmajor, dc);
    if (mode == 2) {
      CK(hipMemsetAsync(dc, 0, NC * 64 * 4, 0));
      hipLaunchKernelGGL(wave_enum_c<R>, dim3(percu * 256), dim3(512), percu == 3 ? 41 * 1024 : 0, 0, dFG, rows, nruns, NW, len2, dbest, wmajor, dc);
    }
  };
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char **argv) {
  const int rows = argc > 1 ? std::atoi(argv[1]) : 4012;
  const int len2 = argc > 2 ? std::atoi(argv[2]) : 2000;
  const int R = argc > 3 ? std::atoi(argv[3]) : 4;
  const int reps = argc > 4 ? std::atoi(argv[4]) : 20;
  const int wmajor = argc > 5 ? std::atoi(argv[5]) : 0;
  const int mode = argc > 6 ? std::atoi(argv[6]) : 0;
  const int percu = argc > 7 ? std::atoi(argv[7]) : 4;
  const int nruns = (len2 + 63) / 64;
  const int stride = 128 * nruns + 64;
  std::vector<int> h((size_t)rows * stride, 0);
  uint32_t x = 12345u;
  auto rnd = [&]() {
    x = x * 1664525u + 1013904223u;
    return (int)(x >> 8) % 200000 - 100000;
  };
  for (int r = 0; r < rows; ++r) {
    int *F = h.data() + (size_t)r * stride, *G = F + 64 * nruns;
    for (int p = 0; p <= len2 - 2; ++p) F[p] = rnd();
    for (int q = 1; q <= len2 - 1; ++q) G[q] = rnd();
    G[0] = G[1];
    for (int q = len2; q < 64 * nruns + 64; ++q) G[q] = G[len2 - 1];
  }
  int *dFG;
  unsigned long long *dbest;
  CK(hipMalloc((void **)&dFG, h.size() * 4));
  CK(hipMemcpy(dFG, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc((void **)&dbest, (size_t)rows * nruns * 8));
  long long *dtrace;
  CK(hipMalloc((void **)&dtrace, (size_t)rows * nruns * 3 * 8));
  CK(hipMemset(dtrace, 0, (size_t)rows * nruns * 3 * 8));
  if (argc > 8) CK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &dtrace, sizeof dtrace));
  int NW = 0;
  float ms = 0;
  if (R == 2) ms = run<2>(dFG, rows, nruns, len2, dbest, reps, &NW, wmajor, mode, percu);
  else if (R == 4) ms = run<4>(dFG, rows, nruns, len2, dbest, reps, &NW, wmajor, mode, percu);
  else if (R == 6) ms = run<6>(dFG, rows, nruns, len2, dbest, reps, &NW, wmajor, mode, percu);
  else ms = run<8>(dFG, rows, nruns, len2, dbest, reps, &NW, wmajor, mode, percu);
  std::vector<unsigned long long> best((size_t)rows * NW);
  CK(hipMemcpy(best.data(), dbest, best.size() * 8, hipMemcpyDeviceToHost));
  if (argc > 8) {
    std::vector<long long> tr((size_t)rows * NW * 3);
    CK(hipMemcpy(tr.data(), dtrace, tr.size() * 8, hipMemcpyDeviceToHost));
    FILE *fp = std::fopen(argv[8], "w");
    for (int u = 0; u < rows * NW; ++u) std::fprintf(fp, "%d %lld %lld %lld\n", u, tr[3 * u], tr[3 * u + 1], tr[3 * u + 2]);
    std::fclose(fp);
  }
  // check a few rows against brute force
  int bad = 0;
  for (int r = 0; r < rows; r += (rows / 7 > 0 ? rows / 7 : 1)) {
    const int *F = h.data() + (size_t)r * stride, *G = F + 64 * nruns;
    long long bv = LLONG_MIN;
    int bp = -1;
    for (int p = 0; p <= len2 - 2; ++p)
      for (int q = p + 1; q <= len2 - 1; ++q)
        if ((long long)F[p] + G[q] > bv) {
          bv = (long long)F[p] + G[q];
          bp = p;
        }
    unsigned long long k = 0;
    for (int w = 0; w < NW; ++w) k = best[(size_t)r * NW + w] > k ? best[(size_t)r * NW + w] : k;
    int gv = (int)((uint32_t)(k >> 32) ^ 0x80000000u), gp = (int)(0xffffffffu - (uint32_t)k);
    if (gv != (int)bv || gp != bp) {
      ++bad;
      std::printf("row %d: gpu (%d,%d) cpu (%lld,%d)\n", r, gv, gp, bv, bp);
    }
  }
  const double cand = (double)rows * ((double)len2 * (len2 - 1) / 2);
  std::printf("rows=%d len2=%d R=%d NW=%d units=%d wmajor=%d mode=%d percu=%d  %.4f ms  %.3e cand/s  frac=%.3f  check=%s\n", rows, len2, R, NW,
              rows * NW, wmajor, mode, percu, ms, cand / (ms * 1e-3), 2 * cand / (ms * 1e-3) / 78.6e12, bad ? "FAIL" : "ok");
  return bad != 0;
}
