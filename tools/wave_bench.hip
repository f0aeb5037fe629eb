// Times psa_enum_wave of libpsa_hip on a synthetic slab (random rows of g), without the
// rest of a run around it.
// Build: hipcc -O3 --offload-arch=gfx950 -I../include -o wave_bench wave_bench.hip \
//        -L../parallel-sequence-alignment_amd -lpsa_hip -Wl,-rpath,'$ORIGIN/../parallel-sequence-alignment_amd'
// Run:   ./wave_bench [offsets] [len2] [enum_ns] [sub_log2] [reps] [lds_bytes] [f32: 1 (default) or 0] [waves per workgroup, default 8]
#include <hip/hip_runtime.h>
#include <limits.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../parallel-sequence-alignment_amd/csrc/psa_kernels.h"

#define CK(x)                                                      \
  do {                                                             \
    hipError_t e_ = (x);                                           \
    if (e_ != hipSuccess) {                                        \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      std::exit(1);                                                \
    }                                                              \
  } while (0)

int main(int argc, char **argv) {
  const int rows = argc > 1 ? std::atoi(argv[1]) : 4012;
  const int len2 = argc > 2 ? std::atoi(argv[2]) : 2000;
  const int ens = argc > 3 ? std::atoi(argv[3]) : 1;
  const int sub_log2 = argc > 4 ? std::atoi(argv[4]) : 3;
  const int reps = argc > 5 ? std::atoi(argv[5]) : 20;
  const int lds = argc > 6 ? std::atoi(argv[6]) : 0;
  const bool f32 = argc > 7 ? std::atoi(argv[7]) != 0 : true;
  const int wpg = argc > 8 ? std::atoi(argv[8]) : 8;
  const int nruns = (len2 + 63) / 64, stride = 64 * nruns + 64;  // a row of g and its run of padding
  const int levels = (nruns + psa::kEnumWaveRuns - 1) / psa::kEnumWaveRuns;
  const int per = std::min(ens << sub_log2, psa::kEnumMaxNs);
  // chunks of `per` offsets, each with per + 1 rows of g (f of an offset = the next row minus its constant)
  std::vector<psa::Chunk> chunks;
  size_t nrows = 0;
  for (int s0 = 0; s0 < rows; s0 += per) {
    const int cnt = std::min(per, rows - s0);
    chunks.push_back(psa::Chunk{0, s0, cnt, (uint32_t)(nrows * (nruns + 1))});
    nrows += (size_t)cnt + 1;
  }
  std::vector<int> h(nrows * stride + 64, 0);
  uint32_t x = 12345u;
  auto rnd = [&]() {
    x = x * 1664525u + 1013904223u;
    return (int)(x >> 8) % 200000 - 100000;
  };
  for (size_t r = 0; r < nrows; ++r) {
    int *G = h.data() + r * stride;
    for (int q = 0; q <= len2 - 1; ++q) G[q] = rnd();
    for (int q = len2; q < 64 * nruns; ++q) G[q] = G[len2 - 1];
    G[64 * nruns] = rnd();  // the row's constant
  }
  psa::SeqDesc d{};
  d.len2 = len2;
  d.s_begin = 0;
  d.s_end = rows;
  d.frac_bits = 0;
  d.enum_ns = ens;
  psa::WaveParams W{};
  W.sub_log2 = sub_log2;
  for (int w = 0; w <= psa::kEnumMaxLevels; ++w) W.level_base[w] = INT_MAX;
  for (int w = 0; w < levels; ++w) W.level_base[w] = w * ((int)chunks.size() << sub_log2);
  W.nunits = levels * ((int)chunks.size() << sub_log2);
  int *dfg;
  psa::Chunk *dch;
  psa::SeqDesc *dd;
  psa::SeqAcc *dacc;
  unsigned long long *dbest;
  CK(hipMalloc((void **)&dfg, h.size() * 4));
  {
    std::vector<int> up(h);
    if (f32)  // the float32 loop reads float32 bit patterns (|values| <= 10^5: exact)
      for (int &v : up) {
        const float fv = (float)v;
        std::memcpy(&v, &fv, 4);
      }
    CK(hipMemcpy(dfg, up.data(), up.size() * 4, hipMemcpyHostToDevice));
  }
  CK(hipMalloc((void **)&dch, chunks.size() * sizeof(psa::Chunk)));
  CK(hipMemcpy(dch, chunks.data(), chunks.size() * sizeof(psa::Chunk), hipMemcpyHostToDevice));
  CK(hipMalloc((void **)&dd, sizeof d));
  CK(hipMemcpy(dd, &d, sizeof d, hipMemcpyHostToDevice));
  CK(hipMalloc((void **)&dacc, sizeof(psa::SeqAcc)));
  CK(hipMemset(dacc, 0, sizeof(psa::SeqAcc)));
  CK(hipMalloc((void **)&dbest, chunks.size() * 8));
  CK(hipMemset(dbest, 0, chunks.size() * 8));
  W.chunks = dch;
  W.desc = dd;
  W.acc = dacc;
  W.chunk_best = dbest;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  CK(psa::launch_enum_wave(dfg, W, f32, wpg, lds, st, nullptr, nullptr));
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(a, st));
  for (int i = 0; i < reps; ++i) CK(psa::launch_enum_wave(dfg, W, f32, wpg, lds, st, nullptr, nullptr));
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  ms /= reps;
  psa::SeqAcc acc;
  CK(hipMemcpy(&acc, dacc, sizeof acc, hipMemcpyDeviceToHost));
  // brute-force check of the overall maximum and its smallest p
  long long bv = LLONG_MIN;
  int bp = -1;
  for (const psa::Chunk &ch : chunks)
    for (int i = 0; i < ch.ns; ++i) {
      const int *G = h.data() + ((size_t)ch.fg_row / (nruns + 1) + i) * stride, *Gn = G + stride;
      int sm = INT_MIN;  // suffix maximum of g
      for (int p = len2 - 2; p >= 0; --p) {
        sm = std::max(sm, G[p + 1]);
        const long long v = (long long)Gn[p] - Gn[64 * nruns] + sm;
        if (v > bv || (v == bv && p < bp)) {
          bv = v;
          bp = p;
        }
      }
    }
  const int gv = (int)((uint32_t)(acc.enum_key >> 32) ^ 0x80000000u), gp = (int)(0xffffffffu - (uint32_t)acc.enum_key);
  const double cand = (double)rows * ((double)len2 * (len2 - 1) / 2);
  std::printf("%s offsets=%d len2=%d enum_ns=%d sub_log2=%d units=%d lds=%d  %.4f ms  %.3e cand/s  frac=%.3f  check=%s\n", f32 ? "f32" : "i32", rows,
              len2, ens, sub_log2, W.nunits, lds, ms, cand / (ms * 1e-3), 2 * cand / (ms * 1e-3) / 78.6432e12,
              (gv == (int)bv && gp == bp) ? "ok" : "FAIL");
  return !(gv == (int)bv && gp == bp);
}
