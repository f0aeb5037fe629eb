// gfx950 enumeration kernels of libpsa_hip: every (offset, mutant) candidate scored in
// fixed point as f_s[p] + g_s[q].
//
// Scored quantity (P/cudaFunctions.cu:205-223, S/main.c:53-61): for offset s and
// deleted 0-based positions p < q of Seq2 (B, length L2) against Seq1 (A),
//   score(s,p,q) = sum_{j<p} T[A[s+j]][B[j]] + sum_{p<j<q} T[A[s+j-1]][B[j]]
//                + sum_{j>q} T[A[s+j-2]][B[j]].
// All three sums run along anti-diagonals delta = s, s-1, s-2 of the L1 x L2
// comparison grid.  With E^delta[x] the exclusive prefix sum of the fixed-point table
// entries along diagonal delta,
//   score = f_s[p] + g_s[q],  f_s[p] = E^s[p] - E^{s-1}[p+1],
//                             g_s[q] = E^{s-1}[q] + E^{s-2}[L2] - E^{s-2}[q+1],
// so every (s,p,q) costs one integer add and one max.  A workgroup owns a run of
// offsets; each lane keeps f for R values of p in registers with one running maximum
// each, g is broadcast from LDS.
//
// Instruction mix (measured on MI355X): v_add_u32 issues at full rate, every integer
// or float min/max at half rate, so two candidates are folded into one v_max3_i32:
// 2 adds + 1 max3 per two candidates.
#include "psa_device.h"

namespace psa {

template <int NT, int R>
struct EnumLds {
  static constexpr int NP = NT * R;  // positions covered (needs NP >= L2 + 1)
  static constexpr int NW = NT / 64;
  // byte offsets inside dynamic LDS
  static constexpr int ring = 0;                     // 3 * NP int32: E^delta for delta mod 3
  static constexpr int g = ring + 3 * NP * 4;        // NP int32 (+ 4 of slack read by the prefetch)
  static constexpr int itab = g + NP * 4 + 16;       // kDevTable int32
  static constexpr int wtot = itab + kDevTable * 4;  // 16 int32
  static constexpr int red = wtot + 16 * 4;          // 16 u64
  static constexpr int misc = red + 16 * 8;          // 4 int32
  static constexpr int awin = misc + 16;             // kEnumMaxNs + NP + 2 bytes, rounded up
  static constexpr int total = awin + ((kEnumMaxNs + NP + 2 + 15) / 16) * 16;
};

// Exclusive prefix sums of one diagonal into ring slot (delta mod 3).
// Thread t sums positions t*R .. t*R+R-1; a wave scan and the wave totals in LDS
// give the base.  Ends with a barrier, so the slot is readable on return.
template <int NT, int R>
__device__ __forceinline__ void scan_diagonal(unsigned char *smem, const int (&bro)[R], int delta, int win0) {
  using L = EnumLds<NT, R>;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int *itab = (const int *)(smem + L::itab);
  int *wtot = (int *)(smem + L::wtot);
  const unsigned char *aw = smem + L::awin + (delta - win0) + tid * R;
  int e[R];
  int run = 0;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    int v = itab[bro[i] + aw[i]];
    e[i] = run;
    run += v;
  }
  int incl = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int nb = __shfl_up(incl, o);
    if (lane >= o) incl += nb;
  }
  if (lane == 63) wtot[w] = incl;
  __syncthreads();
  int base = incl - run;
  for (int i = 0; i < w; ++i) base += wtot[i];
  int slot = delta % 3;
  if (slot < 0) slot += 3;
  int *E = (int *)(smem + L::ring) + slot * L::NP + tid * R;
#pragma unroll
  for (int i = 0; i < R; ++i) E[i] = base + e[i];
  __syncthreads();
}

template <int NT, int R>
__device__ __forceinline__ const int *ring_slot(const unsigned char *smem, int delta) {
  int slot = delta % 3;
  if (slot < 0) slot += 3;
  return (const int *)(smem + EnumLds<NT, R>::ring) + slot * EnumLds<NT, R>::NP;
}

__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// q in [qa, qb) (multiples of 4) against registers 0..K-1, every lane valid:
// the p run of each of those registers lies wholly below qa.
template <int R, int K>
__device__ __forceinline__ void enum_full(const int *gL, int qa, int qb, const int (&f)[R], int (&acc)[R]) {
  if (qa >= qb) return;
  const int4 *g4 = (const int4 *)(gL + qa);
  const int n = (qb - qa) >> 2;
  int4 cur = g4[0];
#pragma unroll 2
  for (int u = 0; u < n; ++u) {
    const int4 nxt = g4[u + 1];  // one group ahead; the last read lands in the slack after g
    // two passes over the registers, so that consecutive max3 instructions never
    // update the same accumulator (a dependent pair would stall the issue)
#pragma unroll
    for (int r = 0; r < K; ++r) acc[r] = max3i(acc[r], f[r] + cur.x, f[r] + cur.y);
#pragma unroll
    for (int r = 0; r < K; ++r) acc[r] = max3i(acc[r], f[r] + cur.z, f[r] + cur.w);
    cur = nxt;
  }
}

template <int NW, int R, int K>
struct FullPhases {
  // registers 0..K-1 are complete for q runs qc in (pc(K-1), pc(K)], pc(r) = r*NW + w
  __device__ static __forceinline__ void run(const int *gL, int w, int qend, const int (&f)[R], int (&acc)[R]) {
    FullPhases<NW, R, K - 1>::run(gL, w, qend, f, acc);
    const int qa = 64 * ((K - 1) * NW + w + 1);
    int qb = (K < R) ? 64 * (K * NW + w + 1) : qend;
    qb = qb < qend ? qb : qend;
    enum_full<R, K>(gL, qa, qb, f, acc);
  }
};
template <int NW, int R>
struct FullPhases<NW, R, 0> {
  __device__ static __forceinline__ void run(const int *, int, int, const int (&)[R], int (&)[R]) {}
};

// The q run that coincides with a register's own p run: lane l holds p = q0 + l, so
// q0 + i pairs only with lanes l < i.  All R registers share the lane mask of step i,
// so their diagonal runs are walked together under one exec mask per step.
template <int NW, int R>
__device__ __forceinline__ void enum_diag(const int *gL, int w, int lane, const int (&f)[R], int (&acc)[R]) {
#pragma unroll 1
  for (int i = 0; i < 64; i += 4) {
    int4 gv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) gv[r] = *(const int4 *)(gL + 64 * (r * NW + w) + i);
    if (lane < i) {
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = max(acc[r], f[r] + gv[r].x);
    }
    if (lane < i + 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = max(acc[r], f[r] + gv[r].y);
    }
    if (lane < i + 2) {
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = max(acc[r], f[r] + gv[r].z);
    }
    if (lane < i + 3) {
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = max(acc[r], f[r] + gv[r].w);
    }
  }
}

// Shared set-up of a chunk: fixed-point table, Seq2 codes of this thread's
// positions, Seq1 window, and the two diagonals older than the first offset.
template <int NT, int R>
__device__ __forceinline__ void enum_chunk_setup(unsigned char *smem, const EnumParams &P, const Chunk &ch,
                                                 const SeqDesc &d, int (&bro)[R], int &win0) {
  using L = EnumLds<NT, R>;
  const int tid = threadIdx.x;
  int *itab = (int *)(smem + L::itab);
  for (int i = tid; i < kDevTable; i += NT) itab[i] = __float2int_rn(ldexpf(P.table_t[i], d.frac_bits));
  const unsigned char *b = P.seq2_pool + d.off2;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    int j = tid * R + i;
    bro[i] = (j < d.len2 ? (int)b[j] : kPadCode) * kRowStride;
  }
  win0 = ch.s0 - 2;
  const int wlen = ch.ns + L::NP + 2;
  unsigned char *aw = smem + L::awin;
  for (int i = tid; i < wlen; i += NT) {
    int64_t idx = (int64_t)win0 + i;
    aw[i] = (idx >= 0 && idx < P.len1) ? P.seq1[idx] : (unsigned char)kPadCode;
  }
  __syncthreads();
  scan_diagonal<NT, R>(smem, bro, ch.s0 - 2, win0);
  scan_diagonal<NT, R>(smem, bro, ch.s0 - 1, win0);
}

// waves per SIMD the register budget is sized for: 8 (at most 64 VGPRs) except for the
// largest variant, whose 32 running maxima and f values per lane need 128
constexpr int enum_waves_per_simd(int nt, int r) { return nt * r > 4096 ? 4 : 8; }

template <int NT, int R>
__global__ __launch_bounds__(NT, enum_waves_per_simd(NT, R)) void psa_enum_main(EnumParams P) {
  using L = EnumLds<NT, R>;
  constexpr int NW = NT / 64;
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = uni(tid >> 6);
  int *gL = (int *)(smem + L::g);
  int *misc = (int *)(smem + L::misc);
  unsigned long long *red = (unsigned long long *)(smem + L::red);

  // Every branch that encloses a barrier tests a scalar (readfirstlane) value, so
  // the compiler keeps the loop wave-uniform and all lanes reach each barrier together.
  if (tid == 0) misc[0] = (int)atomicAdd(P.counter, 1u);
  __syncthreads();
  int c = uni(misc[0]);
  while (c < P.nchunks) {
    const Chunk ch = load_chunk(P.chunks + c);
    const SeqDesc d = load_desc(P.desc + ch.seq);
    const int L2 = d.len2;
    int bro[R];
    int win0;
    enum_chunk_setup<NT, R>(smem, P, ch, d, bro, win0);

    int acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = INT_MIN;
    const int qend = 64 * ((L2 - 1) / 64 + 1);  // q runs cover 1 .. L2-1, padded to whole runs

    for (int s = ch.s0; s < ch.s0 + ch.ns; ++s) {
      scan_diagonal<NT, R>(smem, bro, s, win0);
      const int *Es = ring_slot<NT, R>(smem, s);
      const int *E1 = ring_slot<NT, R>(smem, s - 1);
      const int *E2 = ring_slot<NT, R>(smem, s - 2);
      const int tot2 = E2[L2];
      int f[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int p = r * NT + tid;
        f[r] = (p <= L2 - 2) ? Es[p] - E1[p + 1] : 0;
        int qq = p < 1 ? 1 : (p > L2 - 1 ? L2 - 1 : p);  // positions past the end repeat g[L2-1]
        gL[p] = E1[qq] + tot2 - E2[qq + 1];
      }
      __syncthreads();
      FullPhases<NW, R, R>::run(gL, w, qend, f, acc);
      enum_diag<NW, R>(gL, w, lane, f, acc);
    }

    // best fixed-point score of this chunk and the smallest p attaining it
    int bestv = INT_MIN;
    uint32_t bestp = 0x7fffffffu;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int p = r * NT + tid;
      if (p <= L2 - 2 && acc[r] > bestv) {
        bestv = acc[r];
        bestp = (uint32_t)p;
      }
    }
    unsigned long long key = ((unsigned long long)((uint32_t)bestv ^ 0x80000000u) << 32) | (0xffffffffu - bestp);
    key = wave_max_u64(key);
    if (lane == 0) red[w] = key;
    __syncthreads();
    if (tid == 0) {
      for (int i = 1; i < NW; ++i) key = red[i] > key ? red[i] : key;
      P.chunk_best[c] = key;  // lets the second pass skip chunks that cannot hold the winner
      atomicMax(&P.acc[ch.seq].enum_key, key);
      misc[0] = (int)atomicAdd(P.counter, 1u);  // next chunk
    }
    __syncthreads();
    c = uni(misc[0]);
  }
}

// Second pass: knowing the maximum G and the smallest p* that attains it, find the
// smallest q and then the smallest offset with f_s[p*] + g_s[q] == G.  Only chunks
// whose own (maximum, p) equals the Seq2's can contain such a candidate.
template <int NT, int R>
__global__ __launch_bounds__(NT, enum_waves_per_simd(NT, R)) void psa_enum_locate(EnumParams P) {
  using L = EnumLds<NT, R>;
  constexpr int NW = NT / 64;
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  unsigned long long *red = (unsigned long long *)(smem + L::red);

  for (int c = blockIdx.x; c < P.nchunks; c += gridDim.x) {
    const Chunk ch = load_chunk(P.chunks + c);
    const unsigned long long ek = P.acc[ch.seq].enum_key;
    const unsigned long long ck = P.chunk_best[c];
    if (uni((int)(ek != ck))) continue;
    const SeqDesc d = load_desc(P.desc + ch.seq);
    const int L2 = d.len2;
    int bro[R];
    int win0;
    __syncthreads();
    enum_chunk_setup<NT, R>(smem, P, ch, d, bro, win0);
    const int G = uni((int)((uint32_t)(ek >> 32) ^ 0x80000000u));
    const int pstar = uni((int)(0xffffffffu - (uint32_t)ek));
    unsigned long long best = ~0ull;
    for (int s = ch.s0; s < ch.s0 + ch.ns; ++s) {
      scan_diagonal<NT, R>(smem, bro, s, win0);
      const int *Es = ring_slot<NT, R>(smem, s);
      const int *E1 = ring_slot<NT, R>(smem, s - 1);
      const int *E2 = ring_slot<NT, R>(smem, s - 2);
      const int tot2 = E2[L2];
      const int fstar = Es[pstar] - E1[pstar + 1];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int q = r * NT + tid;
        if (q > pstar && q <= L2 - 1) {
          int gq = E1[q] + tot2 - E2[q + 1];
          if (fstar + gq == G) {
            unsigned long long k = ((unsigned long long)(uint32_t)q << 32) | (uint32_t)s;
            best = k < best ? k : best;
          }
        }
      }
    }
    best = wave_min_u64(best);
    if (lane == 0) red[w] = best;
    __syncthreads();
    if (tid == 0) {
      for (int i = 1; i < NW; ++i) best = red[i] < best ? red[i] : best;
      if (best != ~0ull) atomicMin(&P.acc[ch.seq].enum_loc, best);
    }
  }
}

template <int NT, int R>
static hipError_t launch_enum_t(const EnumParams &P, int grid_main, int grid_locate, hipStream_t st,
                                hipEvent_t ev_begin, hipEvent_t ev_end) {
  using L = EnumLds<NT, R>;
  static bool attr_done = false;
  hipError_t e;
  if (!attr_done) {
    e = hipFuncSetAttribute((const void *)psa_enum_main<NT, R>, hipFuncAttributeMaxDynamicSharedMemorySize, L::total);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)psa_enum_locate<NT, R>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            L::total);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  if (ev_begin && (e = hipEventRecord(ev_begin, st)) != hipSuccess) return e;
  hipLaunchKernelGGL((psa_enum_main<NT, R>), dim3(grid_main), dim3(NT), L::total, st, P);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if (ev_end && (e = hipEventRecord(ev_end, st)) != hipSuccess) return e;
  if ((e = debug_sync("psa_enum_main")) != hipSuccess) return e;
  hipLaunchKernelGGL((psa_enum_locate<NT, R>), dim3(grid_locate), dim3(NT), L::total, st, P);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return debug_sync("psa_enum_locate");
}

int enum_class_of(int len2) {
  for (int c = 0; c < kEnumClasses; ++c)
    if (len2 + 1 <= enum_class_positions(c)) return c;
  return -1;
}

int enum_class_positions(int c) {
  static const int np[kEnumClasses] = {256 * 1, 256 * 2, 512 * 2, 512 * 4, 1024 * 4, 1024 * 8};
  return np[c];
}

int enum_class_lds_bytes(int c) {
  switch (c) {
    case 0: return EnumLds<256, 1>::total;
    case 1: return EnumLds<256, 2>::total;
    case 2: return EnumLds<512, 2>::total;
    case 3: return EnumLds<512, 4>::total;
    case 4: return EnumLds<1024, 4>::total;
    default: return EnumLds<1024, 8>::total;
  }
}

int enum_class_threads(int c) { return c < 2 ? 256 : (c < 4 ? 512 : 1024); }

hipError_t launch_enum(int cls, const EnumParams &P, int grid_main, int grid_locate, hipStream_t st,
                       hipEvent_t ev_begin, hipEvent_t ev_end) {
  switch (cls) {
    case 0: return launch_enum_t<256, 1>(P, grid_main, grid_locate, st, ev_begin, ev_end);
    case 1: return launch_enum_t<256, 2>(P, grid_main, grid_locate, st, ev_begin, ev_end);
    case 2: return launch_enum_t<512, 2>(P, grid_main, grid_locate, st, ev_begin, ev_end);
    case 3: return launch_enum_t<512, 4>(P, grid_main, grid_locate, st, ev_begin, ev_end);
    case 4: return launch_enum_t<1024, 4>(P, grid_main, grid_locate, st, ev_begin, ev_end);
    case 5: return launch_enum_t<1024, 8>(P, grid_main, grid_locate, st, ev_begin, ev_end);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace psa
