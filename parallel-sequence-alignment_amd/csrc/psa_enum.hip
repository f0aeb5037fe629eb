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
// so every (s,p,q) costs one add and one max.
//
// Three kernels:
//   psa_enum_prep    one workgroup (or, in large batches, one wave) per chunk of offsets: prefix
//                    sums along the diagonals, then one row of g per offset into a slab in HBM.  f needs no row of
//                    its own: g_{s+1}[p] = E^s[p] + E^{s-1}[L2] - E^{s-1}[p+1] = f_s[p] + E^{s-1}[L2],
//                    so a chunk of ns offsets stores ns + 1 rows of g, each with its constant
//                    (4 bytes per position and offset, against L2/2 candidates each);
//   psa_enum_wave_mfma  the candidate loop.  One wave per (offsets, 256 positions p): the sums
//                    f[p] + g[q] of 32 x 32 candidates are one v_mfma_f32_32x32x8_bf16_1k over
//                    exact bfloat16 pieces of f and g, the vector ALU takes the maxima (8
//                    v_max3_f32 per 1024 candidates); no LDS, no barriers, so waves are
//                    scheduled one by one and the chip stays full to the last unit;
//   psa_enum_wave    the same loop on the vector ALU alone (f in 4 vector registers per lane, g
//                    through scalar loads): the int32 form and PSA_ENUM_MFMA=0;
//   psa_enum_locate  second pass over the few chunks that hold the maximum: smallest q
//                    and offset for the winning p.
//
// Instruction mix of the vector-ALU form (measured on MI355X): adds issue at full rate, every
// integer or float min/max at half rate, so two candidates are folded into one max3: 2 adds +
// 1 max3 per two candidates.  The float32 pair (v_add_f32, v_max3_f32) issues about a fifth
// faster than the int32 pair, and float32 adds integers of at most 24 bits exactly, so the
// default scale keeps f, g and f + g within 24 bits and the loop runs in float32 -- which is
// also what lets the matrix cores form the sums exactly; the int32 form (30-bit values) is
// kept behind a switch.
#include <stdlib.h>

#include <hip/hip_ext.h>

#include "psa_device.h"

namespace psa {

template <int NT, int R>
struct EnumLds {
  static constexpr int NP = NT * R;  // positions covered (needs NP >= L2 + 1)
  static constexpr int NW = NT / 64;
  // byte offsets inside dynamic LDS
  static constexpr int ring = 0;                     // 3 * NP int32: E^delta for delta mod 3
  static constexpr int itab = ring + 3 * NP * 4;     // kDevTable int32
  static constexpr int wtot = itab + kDevTable * 4;  // 16 int32
  static constexpr int red = wtot + 16 * 4;          // 16 u64
  static constexpr int awin = red + 16 * 8;          // kEnumMaxNs + NP + 2 bytes, rounded up
  // shortlist pass of the locate kernel: 16 wave maxima of f, a counter and a flag, kShortPList positions
  // and as many masks, the maxima of g per run of 64 positions (at most 128) and their suffix maxima
  static constexpr int slist = awin + ((kEnumMaxNs + NP + 2 + 15) / 16) * 16;
  static constexpr int total = slist + (32 + 4 + 2 * kShortPList + 256) * 4;
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

// ---- pre-pass -----------------------------------------------------------------------
// Row of offset s in the slab: 64*nruns g values (q >= L2 repeat g[L2-1]: duplicates of scored
// candidates; q = 0, which no candidate uses, holds the formula's value because f_{s-1}[0] is
// read from it), then one run of padding whose first word is E^{s-2}[L2], the constant that
// turns this row into f of the offset before: f_{s-1}[p] = g_s[p] - E^{s-2}[L2].  A chunk of ns
// offsets s0 .. s0+ns-1 stores the ns + 1 rows s0 .. s0+ns.
// With F32 the values are stored as float32: the scale keeps |f|, |g| and the constants at or
// below 2^24, so the conversion is exact.
//
// Two forms write the same rows.  Both rest on
//   P^d[q] = sum_{j<q} D^d[j]  (exclusive prefix sum along diagonal d),  I^d[q] = P^d[q] + D^d[q],  tot^d = P^d[L2],
//   g_s[q] = P^{s-1}[q] + tot^{s-2} - I^{s-2}[q]:
// everything position q needs of the older diagonal is ONE value, which stays in a register of the lane that
// owns q; no prefix array is kept in LDS.  A run of 64 positions is scanned by six DPP adds.

// ---- pre-pass, one wave per chunk (Seq2 of at most 64 R positions, R <= 32) ------------------
// No barrier at all: a wave owns a whole chunk.  Lane l holds
// the positions 64 k + l of every run k in registers (I^{d-1} and the Seq2 residue's table row), walks diagonal
// after diagonal, run after run -- entry from the table in LDS, inclusive scan of the 64 entries by six
// DPP adds, the run's total to a scalar register that carries the prefix into the next run -- and stores each
// run of a row with one 256-byte instruction.  Nothing is exchanged between waves, so workgroups are single
// waves that start wherever a wave slot is free (beside candidate loops that hold nearly every register, a
// workgroup of 4 to 16 waves waited for a whole CU to drain).  Per 64 values of a row: 2 LDS reads, about 14
// vector instructions, 1 store.  (Workgroups of 8 waves per chunk, beside the candidate loops of a long list,
// took a quarter of the list's whole step: Seq2 of 500 to 2000 characters.)
__device__ __forceinline__ int wave_scan_inclusive(int x) {
  // rows of 16 lanes: Hillis-Steele with the row shifts; lanes shifted in from outside a row contribute 0
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);  // row_shr:8
  // lane 15 of the row before into rows 1 and 3, then lane 31 into rows 2 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31
  return x;
}

template <int R>
struct PrepWaveLds {
  static constexpr int itab = 0;                                   // kDevTable int32: the Seq2's fixed-point table
  static constexpr int awin = kDevTable * 4;                       // Seq1 window: kEnumMaxNs + 64 R + 2 codes
  static constexpr int total = awin + ((kEnumMaxNs + 64 * R + 2 + 15) / 16) * 16;
};

template <int R, bool F32>
__global__ __launch_bounds__(64) void psa_enum_prep_wave(EnumParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  using L = PrepWaveLds<R>;
  const int lane = threadIdx.x;
  const Chunk ch = load_chunk(P.chunks + blockIdx.x);
  const SeqDesc d = load_desc(P.desc + ch.seq);
  const int L2 = d.len2;
  const int nruns = (L2 + 63) >> 6;  // <= R
  const int npos = 64 * nruns;
  int *itab = (int *)(smem + L::itab);
  unsigned char *aw = smem + L::awin;
  for (int i = lane; i < kDevTable; i += 64) itab[i] = __float2int_rn(ldexpf(P.table_t[i], d.frac_bits));
  // byte offset of the table row of this lane's Seq2 residue in run k (positions past the end: the zero row)
  int brow[R];
  const unsigned char *b = P.seq2_pool + d.off2;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int q = 64 * k + lane;
    brow[k] = (q < L2 ? (int)b[q] : kPadCode) * (kRowStride * 4);
  }
  // Seq1 codes met by the diagonals s0 - 2 .. s0 + ns - 1 at positions 0 .. npos - 1
  const int win0 = ch.s0 - 2;
  const int wlen = ch.ns + 2 + npos;
  for (int i = lane; i < wlen; i += 64) {
    const int64_t idx = (int64_t)win0 + i;
    aw[i] = (idx >= 0 && idx < P.len1) ? P.seq1[idx] : (unsigned char)kPadCode;
  }
  __syncthreads();  // (one wave: orders the LDS writes above before the reads below)
  int iprev[R];     // I^{d-1}[q] of this lane's positions
  int tot_prev = 0;
  const int qlast = (L2 - 1) & 63;  // lane of position L2 - 1 in the last run
  int *row = P.fg + (size_t)ch.fg_row * 64;
  // diagonal dd (relative to win0) -> with WRITE the row of offset win0 + dd + 1
  auto walk = [&](int dd, bool write) {
    int sbase = 0;  // P^d at the first position of the run: wave-uniform, carried in a scalar register
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (k < nruns) {
        const int a = aw[dd + 64 * k + lane];
        const int D = *(const int *)((const unsigned char *)itab + brow[k] + 4 * a);
        const int inc = wave_scan_inclusive(D);
        const int I = sbase + inc;
        if (write) {
          int g = I - D + tot_prev - iprev[k];
          if (k == nruns - 1) {  // positions past the end of Seq2 repeat g[L2 - 1]
            const int gl = __builtin_amdgcn_readlane(g, qlast);
            g = lane > qlast ? gl : g;
          }
          row[64 * k + lane] = F32 ? __float_as_int((float)g) : g;
        }
        iprev[k] = I;
        sbase += __builtin_amdgcn_readlane(inc, 63);
      }
    }
    if (write) {
      // one run of padding (rows must not start a power of two apart) whose first word is the row's constant
      if (lane == 0) row[npos] = F32 ? __float_as_int((float)tot_prev) : tot_prev;
      row += npos + 64;
    }
    tot_prev = sbase;
  };
#pragma unroll
  for (int k = 0; k < R; ++k) iprev[k] = 0;
  walk(0, false);                                             // diagonal s0 - 2
  for (int dd = 1; dd <= ch.ns + 1; ++dd) walk(dd, true);     // diagonal s0 - 1 + i gives the row of offset s0 + i
}

// ---- pre-pass, one workgroup per chunk (small batches, where the next candidate loop waits for this
// kernel, and Seq2 too long for one wave's registers) ------------------------------------------------
// Thread t of NT owns the positions k NT + t, k = 0 .. R - 1: wave w of NW owns run k NW + w, and a row is
// stored with one 256-byte instruction per wave and k.  The prefix at the start of a run needs the totals of
// the runs before it, other waves' among them: every wave leaves its runs' totals in LDS, and after the one
// barrier of the diagonal every wave scans the list of totals for itself (at most 128 of them, two registers).
// The list is kept twice, for even and odd diagonals, so a wave may write the next diagonal's totals while
// another still reads this one's.  LDS traffic per value: one byte of the Seq1 window (consecutive across
// the lanes) and one table entry.
template <int NT, int R>
struct PrepBlockLds {
  static constexpr int NW = NT / 64;
  static constexpr int NRUN = R * NW;                               // runs of 64 positions covered (<= 128)
  static constexpr int itab = 0;                                    // kDevTable int32
  static constexpr int wtot = kDevTable * 4;                        // 2 x NRUN int32
  static constexpr int awin = wtot + 2 * NRUN * 4;                  // Seq1 window: kEnumMaxNs + 64 NRUN + 2 codes
  static constexpr int total = awin + ((kEnumMaxNs + 64 * NRUN + 2 + 15) / 16) * 16;
};

template <int NT, int R, bool F32>
__global__ __launch_bounds__(NT) void psa_enum_prep(EnumParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  using L = PrepBlockLds<NT, R>;
  constexpr int NW = L::NW, NRUN = L::NRUN, NREG = (NRUN + 63) / 64;
  static_assert(64 % NW == 0 && NRUN <= 128, "the list of run totals is scanned in at most two registers");
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Chunk ch = load_chunk(P.chunks + blockIdx.x);
  const SeqDesc d = load_desc(P.desc + ch.seq);
  const int L2 = d.len2;
  const int nruns = (L2 + 63) >> 6;  // <= NRUN
  const int npos = 64 * nruns;
  int *itab = (int *)(smem + L::itab);
  int *wtot = (int *)(smem + L::wtot);
  unsigned char *aw = smem + L::awin;
  for (int i = tid; i < kDevTable; i += NT) itab[i] = __float2int_rn(ldexpf(P.table_t[i], d.frac_bits));
  int brow[R];  // byte offset of the table row of this thread's Seq2 residue (positions past the end: the zero row)
  const unsigned char *b = P.seq2_pool + d.off2;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int q = k * NT + tid;
    brow[k] = (q < L2 ? (int)b[q] : kPadCode) * (kRowStride * 4);
  }
  const int win0 = ch.s0 - 2;
  const int wlen = ch.ns + 2 + npos;
  for (int i = tid; i < wlen; i += NT) {
    const int64_t idx = (int64_t)win0 + i;
    aw[i] = (idx >= 0 && idx < P.len1) ? P.seq1[idx] : (unsigned char)kPadCode;
  }
  __syncthreads();
  int iprev[R];
#pragma unroll
  for (int k = 0; k < R; ++k) iprev[k] = 0;
  int tot_prev = 0;
  const int qlast = (L2 - 1) & 63;
  int *row = P.fg + (size_t)ch.fg_row * 64;
  for (int dd = 0; dd <= ch.ns + 1; ++dd) {  // diagonal s0 - 2 + dd; from dd = 1 on it gives the row of offset s0 - 1 + dd
    int *wt = wtot + (dd & 1) * NRUN;
    int D[R], inc[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int rr = k * NW + w;
      D[k] = 0;
      inc[k] = 0;
      if (rr < nruns) {
        const int a = aw[dd + k * NT + tid];
        D[k] = *(const int *)((const unsigned char *)itab + brow[k] + 4 * a);
        inc[k] = wave_scan_inclusive(D[k]);
        if (lane == 63) wt[rr] = inc[k];
      }
    }
    __syncthreads();
    int excl[NREG];
    int tot = 0;
#pragma unroll
    for (int j = 0; j < NREG; ++j) {
      const int x = 64 * j + lane < nruns ? wt[64 * j + lane] : 0;
      const int sc = wave_scan_inclusive(x);
      excl[j] = sc - x + tot;
      tot += __builtin_amdgcn_readlane(sc, 63);
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int rr = k * NW + w;
      if (rr < nruns) {
        const int I = __builtin_amdgcn_readlane(excl[(k * NW) >> 6], rr & 63) + inc[k];
        if (dd > 0) {
          int g = I - D[k] + tot_prev - iprev[k];
          if (rr == nruns - 1) {  // positions past the end of Seq2 repeat g[L2 - 1]
            const int gl = __builtin_amdgcn_readlane(g, qlast);
            g = lane > qlast ? gl : g;
          }
          row[64 * rr + lane] = F32 ? __float_as_int((float)g) : g;
        }
        iprev[k] = I;
      }
    }
    if (dd > 0) {
      // one run of padding (rows must not start a power of two apart) whose first word is the row's constant
      if (tid == 0) row[npos] = F32 ? __float_as_int((float)tot_prev) : tot_prev;
      row += npos + 64;
    }
    tot_prev = tot;
  }
}

// ---- candidate loop -----------------------------------------------------------------
// The two instructions of the candidate loop, pinned: hipcc orders volatile asm
// statements as written, and the order decides the speed here (see wave_full).
// Two arithmetic types.  Int32: v_add_u32 + v_max3_i32.  Float32 holding integers of at
// most 24 bits (every sum exact, so the same candidates win): v_add_f32 + v_max3_f32,
// which this chip issues about a third faster than the integer pair.  Registers are
// declared int in both cases; in the float case they hold float32 bit patterns.
template <bool F32>
struct Cand;
template <>
struct Cand<false> {
  static constexpr int lowest = INT_MIN;
  __device__ static __forceinline__ int add_sv(int g, int f) {
    int t;
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "s"(g), "v"(f));
    return t;
  }
  __device__ static __forceinline__ int add_vv(int g, int f) {
    int t;
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(g), "v"(f));
    return t;
  }
  __device__ static __forceinline__ void max3_acc(int &a, int x, int y) {
    asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
  }
  // max3 under the lane masks l < I0 (first accumulator) and l < I1 (second)
  template <int I0, int I1>
  __device__ static __forceinline__ void max3_masked2(int &a, int x, int y, int &b, int z, int w) {
    asm volatile(
        "s_bfm_b64 exec, %6, 0\n\tv_max3_i32 %0, %0, %2, %3\n\ts_bfm_b64 exec, %7, 0\n\t"
        "v_max3_i32 %1, %1, %4, %5\n\ts_mov_b64 exec, -1"
        : "+v"(a), "+v"(b)
        : "v"(x), "v"(y), "v"(z), "v"(w), "n"(I0), "n"(I1));
  }
  template <int I0>
  __device__ static __forceinline__ void max3_masked1(int &a, int x, int y) {
    asm volatile("s_bfm_b64 exec, %3, 0\n\tv_max3_i32 %0, %0, %1, %2\n\ts_mov_b64 exec, -1"
                 : "+v"(a)
                 : "v"(x), "v"(y), "n"(I0));
  }
  __device__ static __forceinline__ int add(int a, int b) { return a + b; }
  __device__ static __forceinline__ int sub(int a, int b) { return a - b; }
  __device__ static __forceinline__ int max2(int a, int b) { return max(a, b); }
  __device__ static __forceinline__ int to_int(int a) { return a; }
};
template <>
struct Cand<true> {
  static constexpr int lowest = (int)0xce800000u;  // -2^30 as float32: below every score, converts to an int
  __device__ static __forceinline__ int add_sv(int g, int f) {
    int t;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(t) : "s"(g), "v"(f));
    return t;
  }
  __device__ static __forceinline__ int add_vv(int g, int f) {
    int t;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(g), "v"(f));
    return t;
  }
  __device__ static __forceinline__ void max3_acc(int &a, int x, int y) {
    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));
  }
  template <int I0, int I1>
  __device__ static __forceinline__ void max3_masked2(int &a, int x, int y, int &b, int z, int w) {
    asm volatile(
        "s_bfm_b64 exec, %6, 0\n\tv_max3_f32 %0, %0, %2, %3\n\ts_bfm_b64 exec, %7, 0\n\t"
        "v_max3_f32 %1, %1, %4, %5\n\ts_mov_b64 exec, -1"
        : "+v"(a), "+v"(b)
        : "v"(x), "v"(y), "v"(z), "v"(w), "n"(I0), "n"(I1));
  }
  template <int I0>
  __device__ static __forceinline__ void max3_masked1(int &a, int x, int y) {
    asm volatile("s_bfm_b64 exec, %3, 0\n\tv_max3_f32 %0, %0, %1, %2\n\ts_mov_b64 exec, -1"
                 : "+v"(a)
                 : "v"(x), "v"(y), "n"(I0));
  }
  __device__ static __forceinline__ int add(int a, int b) {
    return __float_as_int(__int_as_float(a) + __int_as_float(b));
  }
  __device__ static __forceinline__ int sub(int a, int b) {
    return __float_as_int(__int_as_float(a) - __int_as_float(b));
  }
  __device__ static __forceinline__ int max2(int a, int b) {
    return __float_as_int(fmaxf(__int_as_float(a), __int_as_float(b)));
  }
  __device__ static __forceinline__ int to_int(int a) { return (int)__int_as_float(a); }
};

// 16 values of g (scalar registers) against registers 0..K-1, every lane valid: four
// blocks of 4 g values; the sums of a block are consumed by the max3 two blocks later
// (A0 A1 M0 A2 M1 A3 M2 M3).  Left to itself hipcc sometimes emits add, add, max3 on two
// temporaries, where every max3 waits for the adds right before it: 15 % slower.  Two
// accumulators per register keep consecutive max3 on different registers.  With all R
// registers in play a g value is first copied to a vector register (one v_mov_b32 for R
// adds): adds with a scalar operand issue about 0.6 cycles slower than vector-only ones.
template <int R, int K, bool F32>
__device__ __forceinline__ void wave_step16(const int *g, const int (&f)[R], int (&acc)[R][2]) {
  int t[4][K][4];
  auto sums = [&](int b) {
    if (K >= R) {
      int vg[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(vg[i]) : "s"(g[4 * b + i]));
#pragma unroll
      for (int r = 0; r < K; ++r) {
        t[b][r][0] = Cand<F32>::add_vv(vg[0], f[r]);
        t[b][r][1] = Cand<F32>::add_vv(vg[1], f[r]);
      }
#pragma unroll
      for (int r = 0; r < K; ++r) {
        t[b][r][2] = Cand<F32>::add_vv(vg[2], f[r]);
        t[b][r][3] = Cand<F32>::add_vv(vg[3], f[r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < K; ++r) {
        t[b][r][0] = Cand<F32>::add_sv(g[4 * b], f[r]);
        t[b][r][1] = Cand<F32>::add_sv(g[4 * b + 1], f[r]);
      }
#pragma unroll
      for (int r = 0; r < K; ++r) {
        t[b][r][2] = Cand<F32>::add_sv(g[4 * b + 2], f[r]);
        t[b][r][3] = Cand<F32>::add_sv(g[4 * b + 3], f[r]);
      }
    }
  };
  auto maxima = [&](int b) {
#pragma unroll
    for (int r = 0; r < K; ++r) Cand<F32>::max3_acc(acc[r][0], t[b][r][0], t[b][r][1]);
#pragma unroll
    for (int r = 0; r < K; ++r) Cand<F32>::max3_acc(acc[r][1], t[b][r][2], t[b][r][3]);
  };
  sums(0);
  sums(1);
  maxima(0);
  sums(2);
  maxima(1);
  sums(3);
  maxima(2);
  maxima(3);
}

// The long phase's step with the scalar load of the step after it issued in its shadow: the 16
// values of this step are waited for at their first use (the copies of the first block), the
// next 16 are requested right behind that (scalar loads return out of order, so a wait always
// covers every load in flight: the request must follow the wait, which the compiler barrier
// pins), and the rest of the step covers their latency.
template <int R, bool F32>
__device__ __forceinline__ void wave_step16_next(const int *g, int *gn, const int *__restrict__ Gn,
                                                 const int (&f)[R], int (&acc)[R][2]) {
  int t[4][R][4];
  auto sums = [&](int b) {
    int vg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(vg[i]) : "s"(g[4 * b + i]));
    if (b == 0) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) gn[i] = Gn[i];
      asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      t[b][r][0] = Cand<F32>::add_vv(vg[0], f[r]);
      t[b][r][1] = Cand<F32>::add_vv(vg[1], f[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      t[b][r][2] = Cand<F32>::add_vv(vg[2], f[r]);
      t[b][r][3] = Cand<F32>::add_vv(vg[3], f[r]);
    }
  };
  auto maxima = [&](int b) {
#pragma unroll
    for (int r = 0; r < R; ++r) Cand<F32>::max3_acc(acc[r][0], t[b][r][0], t[b][r][1]);
#pragma unroll
    for (int r = 0; r < R; ++r) Cand<F32>::max3_acc(acc[r][1], t[b][r][2], t[b][r][3]);
  };
  sums(0);
  sums(1);
  maxima(0);
  sums(2);
  maxima(1);
  sums(3);
  maxima(2);
  maxima(3);
}

// q in [qa, qb) (multiples of 16) against all R registers.  G is wave-uniform, so the 16
// values of a step come in with one scalar load, requested one step ahead.  The last request
// of a row may reach 16 values past qb: into the row's padding at most.
template <int R, bool F32>
__device__ __forceinline__ void wave_full(const int *__restrict__ G, int qa, int qb, const int (&f)[R],
                                          int (&acc)[R][2]) {
  if (qa >= qb) return;
  int ga[16], gb[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) ga[i] = G[qa + i];
  int q = qa;
  for (; q + 32 <= qb; q += 32) {
    wave_step16_next<R, F32>(ga, gb, G + q + 16, f, acc);
    wave_step16_next<R, F32>(gb, ga, G + q + 32, f, acc);
  }
  if (q < qb) wave_step16<R, R, F32>(ga, f, acc);
}

// ---- candidate sums on the matrix cores --------------------------------------------------
// f[p] + g[q] for 32 q x 32 p is the product of a 32 x K and a K x 32 matrix whose entries are
// pieces of the values and ones: sum_k A[q][k] B[k][p] with A = (g_hi, g_mid, 0, g_lo, 1, 1, 0, 1),
// B = (1, 1, 0, 1, f_hi, f_mid, 0, f_lo).  f and g are integers of at most 25 significant bits held in
// float32, so each splits exactly into three bfloat16 values (8 significant bits each): hi = the
// value with its low 16 bits cleared, mid likewise of the remainder, lo the rest.  Every product
// is exact, every partial sum of pieces is an integer no larger than the larger of |f|, |g| and
// |f + g| <= 2^24, so the float32 accumulation inside the matrix instruction is exact in any
// order and the tile holds the same 1024 integers the add instructions would give.
// The instruction is v_mfma_f32_32x32x8_bf16_1k: K = 8 is all the product needs, its operands are
// two registers each instead of four, it occupies the matrix core for 32 cycles like the K = 16
// form but holds the vector issue port for fewer of them (measured with the 8 v_max3_f32 that
// reduce a tile, four waves per SIMD: 39.5 against 43.4 cycles per tile; the vector ALU alone,
// 16 v_add_f32 + 8 v_max3_f32 for the same 1024 candidates, takes 66).
//
// Operand layout (lane l, r = l & 31, h = l >> 5): A[row r][k = 4h + j] and B[k = 4h + j][col r] for
// element j = 0..3 (two per operand register); result register v of lane l is row
// (v & 3) + 8 (v >> 2) + 4h, column r.  Columns are positions p (a lane keeps the running maximum
// of its own p over the rows, i.e. over q), rows are positions q.  So
//   A: lanes  0..31 (k = 0..3) = (g_hi | g_mid), (0 | g_lo) of row r;   lanes 32..63 (k = 4..7) = (1 | 1), (0 | 1)
//   B: lanes  0..31            = (1 | 1), (0 | 1);   lanes 32..63 = (f_hi | f_mid), (0 | f_lo) of column r
// (the low piece sits in the upper half of its register: it has at most 8 significant bits, so the float32 itself,
// whose lower 16 bits are zero, IS the pair (0 | lo) -- no shift)
// With the packed pieces of 64 consecutive values in two registers (lane = position),
// v_permlane32_swap_b32 against a register of ones yields the operand registers of BOTH blocks of
// 32 at once: it exchanges the upper half of its first register with the lower half of its second.
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) int i32x2;
constexpr int kOnesBf16 = 0x3f803f80;   // (1.0, 1.0) as two bfloat16
constexpr int kOneLoBf16 = 0x3f800000;  // (0.0, 1.0)

// x = hi + mid + lo exactly; p0 = (hi | mid << 16), p1 = (0 | lo << 16) as packed bfloat16
__device__ __forceinline__ void split_bf16x3(int xbits, int &p0, int &p1) {
  const float x = __int_as_float(xbits);
  const int hi = xbits & (int)0xffff0000;
  const float rem = x - __int_as_float(hi);
  const int mid = __float_as_int(rem) & (int)0xffff0000;
  const float lo = rem - __int_as_float(mid);
  p0 = (int)__builtin_amdgcn_perm((unsigned)mid, (unsigned)hi, 0x07060302u);
  p1 = __float_as_int(lo);
}

// B operands of the wave's 8 blocks of 32 positions p from its 4 registers of f (lane = position):
// ones in the lower lanes, the pieces of the block's positions in the upper lanes
__device__ __forceinline__ void mfma_build_b(const int (&f)[4], i32x2 (&B)[8]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int p0, p1;
    split_bf16x3(f[r], p0, p1);
    // [ones.low | p.low] = block of positions 64 r + 0..31, [ones.high | p.high] = positions 64 r + 32..63
    const auto s0 = __builtin_amdgcn_permlane32_swap((unsigned)kOnesBf16, (unsigned)p0, false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap((unsigned)kOneLoBf16, (unsigned)p1, false, false);
    B[2 * r] = i32x2{(int)s0[0], (int)s1[0]};
    B[2 * r + 1] = i32x2{(int)s0[1], (int)s1[1]};
  }
}

// one tile: 32 q (operand a) x 32 p (operand b) on top of c (zero, or the diagonal tile's mask);
// running maxima of the lane's column
__device__ __forceinline__ f32x16 mfma_sums(const i32x2 &a, const i32x2 &b, const f32x16 &c) {
  return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(bf16x4, a), __builtin_bit_cast(bf16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ void mfma_reduce(const f32x16 &d, float &m) {
  // a tree over the 16 results, the running maximum joining last: 8 v_max3_f32
  const float t0 = fmaxf(fmaxf(d[0], d[1]), d[2]);
  const float t1 = fmaxf(fmaxf(d[3], d[4]), d[5]);
  const float t2 = fmaxf(fmaxf(d[6], d[7]), d[8]);
  const float t3 = fmaxf(fmaxf(d[9], d[10]), d[11]);
  const float t4 = fmaxf(fmaxf(d[12], d[13]), d[14]);
  const float u0 = fmaxf(fmaxf(t0, t1), t2);
  const float u1 = fmaxf(fmaxf(t3, t4), d[15]);
  m = fmaxf(fmaxf(m, u0), u1);
}
// N tiles that share the operand a (32 q) against blocks 0..N-1 of p, the last one on top of c_last:
// the matrix core forms the sums of tile t + 1 while the vector ALU takes the maxima of tile t
// (two result tiles in registers; the order is pinned, the compiler otherwise puts each tile's
// maxima right behind its own matrix instruction and the wave idles through its latency).
template <int N>
__device__ __forceinline__ void mfma_tiles(const i32x2 &a, const i32x2 (&B)[8], const f32x16 &c_last, float (&m)[8]) {
  const f32x16 z = {};
  f32x16 d = mfma_sums(a, B[0], N == 1 ? c_last : z);
#pragma unroll
  for (int pb = 0; pb < N; ++pb) {
    f32x16 dn = d;
    if (pb + 1 < N) {
      dn = mfma_sums(a, B[pb + 1], pb + 2 == N ? c_last : z);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_reduce(d, m[pb]);
    __builtin_amdgcn_sched_barrier(0);
    d = dn;
  }
}

// A operands of the two blocks of 32 consecutive q whose g values the lanes hold (lane = position):
// the pieces of the block's positions in the lower lanes, ones in the upper lanes
__device__ __forceinline__ void mfma_build_a(int g, i32x2 &a0, i32x2 &a1) {
  int p0, p1;
  split_bf16x3(g, p0, p1);
  const auto s0 = __builtin_amdgcn_permlane32_swap((unsigned)p0, (unsigned)kOnesBf16, false, false);
  const auto s1 = __builtin_amdgcn_permlane32_swap((unsigned)p1, (unsigned)kOneLoBf16, false, false);
  a0 = i32x2{(int)s0[0], (int)s1[0]};  // positions 0..31 of the 64
  a1 = i32x2{(int)s0[1], (int)s1[1]};  // positions 32..63
}

// The block of 32 q with index RB (0..7) inside the level's own 256 positions: every block of p
// below it in full, and block RB itself along the diagonal, where only q > p exists: the mask
// tile adds -2^30 to the other entries.
template <int RB>
__device__ __forceinline__ void mfma_own_block(const i32x2 &a, const i32x2 (&B)[8], const f32x16 &mask, float (&m)[8]) {
  mfma_tiles<RB + 1>(a, B, mask, m);
}

// The 16 tiles of two blocks of 32 q against all 8 blocks of p as ONE pipeline (the second block's
// first tile is formed while the first block's last is reduced: one fill and one drain per 16 tiles).
__device__ __forceinline__ void mfma_tiles16(const i32x2 &a0, const i32x2 &a1, const i32x2 (&B)[8], float (&m)[8]) {
  const f32x16 z = {};
  f32x16 d = mfma_sums(a0, B[0], z);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    f32x16 dn = d;
    if (t + 1 < 16) {
      dn = mfma_sums(t + 1 < 8 ? a0 : a1, B[(t + 1) & 7], z);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_reduce(d, m[t & 7]);
    __builtin_amdgcn_sched_barrier(0);
    d = dn;
  }
}

// A level's own 256 x 256 positions when all of them exist: its 36 tiles (block rb of q against
// blocks 0..rb of p, the last one of each along the diagonal) as one pipeline.
__device__ __forceinline__ constexpr int own_rb(int t) {
  int rb = 0;
  while ((rb + 1) * (rb + 2) / 2 <= t) ++rb;
  return rb;
}
__device__ __forceinline__ constexpr int own_pb(int t) { return t - own_rb(t) * (own_rb(t) + 1) / 2; }
__device__ __forceinline__ void mfma_own_level(const int (&g)[4], const i32x2 (&B)[8], const f32x16 &mask, float (&m)[8]) {
  const f32x16 z = {};
  i32x2 a0, a1;  // operands of the two blocks of the run of 64 q being scored, built when its first tile is issued
  mfma_build_a(g[0], a0, a1);
  f32x16 d = mfma_sums(a0, B[0], mask);
#pragma unroll
  for (int t = 0; t < 36; ++t) {
    f32x16 dn = d;
    if (t + 1 < 36) {
      const int rb = own_rb(t + 1), pb = own_pb(t + 1);
      if (pb == 0 && (rb & 1) == 0) mfma_build_a(g[rb >> 1], a0, a1);
      dn = mfma_sums((rb & 1) ? a1 : a0, B[pb], pb == rb ? mask : z);
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_reduce(d, m[own_pb(t)]);
    __builtin_amdgcn_sched_barrier(0);
    d = dn;
  }
}

// unit number -> (chunk c, sub-run j, level w); false if the unit is empty
__device__ __forceinline__ bool decode_unit(const WaveParams &P, int u, int &c, int &j, int &w) {
  if (P.band > 0) {
    const int per_band = P.ncols8 * P.band;
    const int b = u / per_band, r = u - b * per_band;
    const int g = r / (8 * P.band), rr = r - g * 8 * P.band;
    w = b * P.band + (rr >> 3);
    const int col = (g << 3) + (rr & 7);
    if (col >= P.ncols || w >= P.nlev) return false;
    c = col >> P.sub_log2;
    j = col & ((1 << P.sub_log2) - 1);
    return true;
  }
  w = 0;
#pragma unroll
  for (int i = 1; i < kEnumMaxLevels; ++i) w += (u >= P.level_base[i]) ? 1 : 0;
  const int mu = u - P.level_base[w];
  c = mu >> P.sub_log2;
  j = mu & ((1 << P.sub_log2) - 1);
  return true;
}

// One unit with every candidate's sum formed on the matrix cores: sub-run j of chunk c at level w
// (positions p in [256 w, 256 w + 256) as 8 blocks of 32; q from 256 w to the end of the row in
// blocks of 32).  The row of an offset is read through per-lane loads of 64 consecutive values,
// requested three steps ahead; f of an offset is the next offset's row minus that row's constant.
template <bool STAMP>
__device__ __forceinline__ void wave_unit_mfma(const int *__restrict__ fg, const WaveParams &P, int u, int lane) {
  // diagnostic build only: shader cycles and 100 MHz ticks this unit took, summed over all units,
  // give the clock the chip held while the kernel ran
  unsigned long long stamp_c = 0, stamp_r = 0;
  if constexpr (STAMP) {
    stamp_c = __builtin_amdgcn_s_memtime();
    stamp_r = __builtin_amdgcn_s_memrealtime();
  }
  int c, j, w;
  if (!decode_unit(P, u, c, j, w)) return;
  const Chunk ch = load_chunk(P.chunks + c);
  const SeqDesc d = load_desc(P.desc + ch.seq);
  const int r0 = j * d.enum_ns;
  if (r0 >= ch.ns) return;
  const int r1 = min(r0 + d.enum_ns, ch.ns);
  const int L2 = d.len2;
  const int nruns = (L2 + 63) >> 6;
  if (4 * w >= nruns) return;    // (banded numbering: this Seq2 has fewer levels than the launch's longest)
  const int qend = 64 * nruns;   // row length
  const int pc0 = 4 * w;         // first run of 64 positions p of this level
  const int q00 = 64 * pc0;      // first position of the level
  const int rowlen = qend + 64;  // a row of g and its run of padding
  const int *__restrict__ G = fg + ((size_t)ch.fg_row + (size_t)r0 * (nruns + 1)) * 64;  // row of offset r0
  const __amdgpu_buffer_rsrc_t rows = __builtin_amdgcn_make_buffer_rsrc((void *)G, 0, -1, 0x00020000);
  // Every per-lane load of the unit is the lane's byte offset inside a run of 64 values (ONE vector register) plus a
  // scalar offset: the row, the run and the clamps are wave-uniform and cost scalar instructions only.
  const int vl = 4 * lane;
  // Register r of f holds run pc0 + r of p; past the last run it repeats that run (its maxima are dropped below).
  auto sb = [&](int r) { return 256 * min(pc0 + r, nruns - 1); };
  // the rows' constants are read through the scalar cache, from an address the compiler knows to be wave-uniform
  const char *const Gs = (const char *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uintptr_t)G >> 32)) << 32) |
                                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)G));
  // mask of a diagonal tile: result register v of this lane is row (v & 3) + 8 (v >> 2) + 4 (lane >> 5), column lane & 31
  f32x16 mask;
#pragma unroll
  for (int v = 0; v < 16; ++v)
    mask[v] = ((v & 3) + 8 * (v >> 2) + 4 * (lane >> 5)) > (lane & 31) ? 0.0f : __int_as_float(Cand<true>::lowest);
  float m[8];
#pragma unroll
  for (int pb = 0; pb < 8; ++pb) m[pb] = __int_as_float(Cand<true>::lowest);
  int soff = 0;  // byte offset of the current row
  // Loads are issued unconditionally, at clamped addresses, and in one place per offset: the
  // compiler then counts them exactly, and no wait covers a load younger than one offset's work.
  // (With loads under conditions it waited for everything outstanding, including the requests
  // it had just issued for the next offset: a full memory latency per offset and level.)
  //   fn[r]  next offset's f (its row is the one after this offset's), run pc0 + r of p
  //   gn[k]  next offset's g for the level's own 256 positions, run pc0 + k
  const int step = 4 * rowlen;  // bytes from one offset's row to the next
  auto sq = [&](int q) { return 4 * min(q, qend); };  // past the row: its run of padding
  int fn[4], gn[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    fn[r] = __builtin_amdgcn_raw_buffer_load_b32(rows, vl, soff + step + sb(r), 0);
    gn[r] = __builtin_amdgcn_raw_buffer_load_b32(rows, vl, soff + sq(q00 + 64 * r), 0);
  }
  int tn = *(const int *)(Gs + (step + 4 * qend));  // the constant of the next row
#ifdef PSA_TRACE
  unsigned long long tr_top = 0, tr_own = 0, tr_long = 0, tr_t;
#define PSA_TR(acc) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc += n_ - tr_t; tr_t = n_; }
  tr_t = __builtin_amdgcn_s_memtime();
#else
#define PSA_TR(acc)
#endif
  for (int i = r0; i < r1; ++i) {
    i32x2 B[8];
    {
      int f[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) f[r] = Cand<true>::sub(fn[r], tn);
      mfma_build_b(f, B);
    }
    int gown[4];  // this offset's g for the level's own four runs (gn is requested anew below)
#pragma unroll
    for (int k = 0; k < 4; ++k) gown[k] = gn[k];
    const int grow = soff;  // this offset's row of g
    const int adv = i + 1 < r1 ? step : 0;  // the last offset of the unit requests its own rows again
    soff += step;
    {
      const int nrow = grow + adv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        fn[r] = __builtin_amdgcn_raw_buffer_load_b32(rows, vl, nrow + step + sb(r), 0);
        gn[r] = __builtin_amdgcn_raw_buffer_load_b32(rows, vl, nrow + sq(q00 + 64 * r), 0);
      }
      tn = *(const int *)(Gs + (nrow + step + 4 * qend));
    }
    int q = q00 + 256;
    // the first three steps beyond the level's own positions, requested before those are scored
    int g[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) g[k] = __builtin_amdgcn_raw_buffer_load_b32(rows, vl, grow + sq(q + 64 * k), 0);  // (used ~36 tiles later)
    // the level's own 256 positions: blocks 0..7 of q against the blocks of p below and at them
#define PSA_OWN_STEP(K)                                                          \
    if (q00 + 64 * K < qend) {                                                     \
      i32x2 a0, a1;                                                                \
      mfma_build_a(gown[K], a0, a1);                                               \
      mfma_own_block<2 * K>(a0, B, mask, m);                                       \
      if (q00 + 64 * K + 32 < L2) mfma_own_block<2 * K + 1>(a1, B, mask, m);       \
    }
    PSA_TR(tr_top)
    if (q00 + 224 < L2) {  // all eight blocks of q exist
      mfma_own_level(gown, B, mask, m);
    } else {
      PSA_OWN_STEP(0)
      PSA_OWN_STEP(1)
      PSA_OWN_STEP(2)
      PSA_OWN_STEP(3)
    }
#undef PSA_OWN_STEP
    PSA_TR(tr_own)
    // beyond them every (p, q) pair is a candidate; the row is requested three steps ahead into
    // four registers that take turns (no copies, and a wait only ever covers a load three steps old)
    const f32x16 z = {};
    // four whole steps at a time while the row has them (no condition inside: the running maxima stay in their
    // registers; with the row's last, partial steps in the same loop the two ways through a step were merged by
    // eight register copies per step) ...
    for (; q + 224 < L2; q += 256) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int qn = q + 64 * k + 192 < qend ? q + 64 * k + 192 : q + 64 * k;
        g[(k + 3) & 3] = __builtin_amdgcn_raw_buffer_load_b32(rows, vl, grow + 4 * qn, 0);
        i32x2 a0, a1;
        mfma_build_a(g[k], a0, a1);
        mfma_tiles16(a0, a1, B, m);
      }
    }
    // ... then the row's last one to four steps, the last of which may be half a step
    if (q < qend) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // (past the row's end the request repeats this step's own run, which the wave read three
        // steps ago and the CU's cache still holds: the loads stay unconditional, so the compiler
        // counts them exactly, and none is still in flight when the next offset waits for its rows)
        const int qn = q + 64 * k + 192 < qend ? q + 64 * k + 192 : q + 64 * k;
        g[(k + 3) & 3] = __builtin_amdgcn_raw_buffer_load_b32(rows, vl, grow + 4 * qn, 0);
        if (q + 64 * k < qend) {
          i32x2 a0, a1;
          mfma_build_a(g[k], a0, a1);
          if (q + 64 * k + 32 < L2) mfma_tiles16(a0, a1, B, m);
          else mfma_tiles<8>(a0, B, z, m);
        }
      }
    }
    PSA_TR(tr_long)
  }
#undef PSA_TR
#ifdef PSA_TRACE
  if (lane == 0 && (u % 509) == 0)
    printf("unit %d level %d L2 %d offsets %d: per offset top %llu own %llu long %llu cycles\n", u, w, L2, r1 - r0,
           tr_top / (r1 - r0), tr_own / (r1 - r0), tr_long / (r1 - r0));
#endif
  // best fixed-point score of this unit and the smallest p attaining it; lanes l and l + 32 hold
  // maxima of the same column over different rows and enter the reduction as they are
  int bestv = INT_MIN;
  uint32_t bestp = 0x7fffffffu;
#pragma unroll
  for (int pb = 0; pb < 8; ++pb) {
    const int p = q00 + 32 * pb + (lane & 31);
    const int a = (int)m[pb];
    if (p <= L2 - 2 && a > bestv) {
      bestv = a;
      bestp = (uint32_t)p;
    }
  }
  unsigned long long key = ((unsigned long long)((uint32_t)bestv ^ 0x80000000u) << 32) | (0xffffffffu - bestp);
  key = wave_max_u64(key);
  if (lane == 0) {
    // a stale read can only be too small, which costs an atomic, never a result
    if (key > P.chunk_best[c]) atomicMax(&P.chunk_best[c], key);  // lets the locate pass skip chunks
    if (key > P.acc[ch.seq].enum_key) atomicMax(&P.acc[ch.seq].enum_key, key);
  }
  if constexpr (STAMP) {
    stamp_c = __builtin_amdgcn_s_memtime() - stamp_c;
    stamp_r = __builtin_amdgcn_s_memrealtime() - stamp_r;
    if (lane == 0) {
      // 64 sets of counters, one 64-byte line each: thousands of waves adding to one address queue up
      // behind each other and the kernel no longer runs at the pace (and clock) it is measured for
      unsigned long long *slot = P.clock_probe + 8 * (u & 63);
      atomicAdd(&slot[0], stamp_c);
      atomicAdd(&slot[1], stamp_r);
      atomicAdd(&slot[2], 1ull);
    }
  }
}

// The q run that coincides with a register's own p run: lane l holds p = q0 + l, so
// q0 + i pairs only with lanes l < i.  Steps are taken two at a time: for even i the
// lanes l < i take q0 + i and q0 + i + 1 in one max3; what that leaves out, lane i with
// q0 + i + 1, is the pair (p, p + 1) of every even lane and is added by the caller from a
// per-lane load.  The sums are formed in every lane; only the max3 runs under the lane
// mask, which is written to exec and restored inside one asm statement (the wave is
// whole here: a unit is wave-uniform work).  One call takes steps I .. I + 3 (g[0..3]).
template <int I, bool F32>
__device__ __forceinline__ void diag_group(const int *g, int f, int (&acc)[2]) {
  if constexpr (I == 0) {  // steps 0 and 1 pair with no lane
    const int t2 = Cand<F32>::add_sv(g[2], f), t3 = Cand<F32>::add_sv(g[3], f);
    Cand<F32>::template max3_masked1<2>(acc[1], t2, t3);
  } else {
    const int t0 = Cand<F32>::add_sv(g[0], f), t1 = Cand<F32>::add_sv(g[1], f);
    const int t2 = Cand<F32>::add_sv(g[2], f), t3 = Cand<F32>::add_sv(g[3], f);
    Cand<F32>::template max3_masked2<I, I + 2>(acc[0], t0, t1, acc[1], t2, t3);
  }
}

// The q runs of the level's own R runs of p: run pc0 + J pairs with every lane of
// registers 0..J-1 and with register J along the diagonal, 16 values of g at a time (one
// scalar load each, as in the long phase: holding all 64 values of a run in scalar registers
// left too few for the loop's own scalars, which hipcc then parked in vector-register lanes).
template <int R, int J, int C, bool F32>
__device__ __forceinline__ void own_chunk(const int *g, int *gn, const int *__restrict__ Gn, const int (&f)[R],
                                          int (&acc)[R][2]) {
  // the wait for this chunk's values sits here, and the request for the next chunk's right behind it
  asm volatile("" ::"s"(g[0]));
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 16; ++i) gn[i] = Gn[i];
  asm volatile("" ::: "memory");
  if constexpr (J > 0) wave_step16<R, J, F32>(g, f, acc);
  diag_group<16 * C, F32>(g, f[J], acc[J]);
  diag_group<16 * C + 4, F32>(g + 4, f[J], acc[J]);
  diag_group<16 * C + 8, F32>(g + 8, f[J], acc[J]);
  diag_group<16 * C + 12, F32>(g + 12, f[J], acc[J]);
}
// ga holds the first 16 values of run pc0 + J on entry and those of run pc0 + J + 1 on return
// (the last request of a level reaches into the run after its own runs: the row's padding at most).
template <int R, int J, bool F32>
struct WaveOwn {
  __device__ static __forceinline__ void run(const int *__restrict__ G, int pc0, int nruns, const int (&f)[R],
                                             int (&acc)[R][2], int (&ga)[16], int (&gb)[16]) {
    WaveOwn<R, J - 1, F32>::run(G, pc0, nruns, f, acc, ga, gb);
    if (pc0 + J < nruns) {
      const int *__restrict__ Gj = G + 64 * (pc0 + J);
      own_chunk<R, J, 0, F32>(ga, gb, Gj + 16, f, acc);
      own_chunk<R, J, 1, F32>(gb, ga, Gj + 32, f, acc);
      own_chunk<R, J, 2, F32>(ga, gb, Gj + 48, f, acc);
      own_chunk<R, J, 3, F32>(gb, ga, Gj + 64, f, acc);
    }
  }
};
template <int R, bool F32>
struct WaveOwn<R, -1, F32> {
  __device__ static __forceinline__ void run(const int *__restrict__, int, int, const int (&)[R], int (&)[R][2],
                                             int (&)[16], int (&)[16]) {}
};

// One unit: sub-run j of chunk c at level w (positions p in [256 w, 256 w + 256)).
template <int R, bool F32>
__device__ __forceinline__ void wave_unit(const int *__restrict__ fg, const WaveParams &P, int u, int lane) {
  int c, j, w;
  if (!decode_unit(P, u, c, j, w)) return;
  const Chunk ch = load_chunk(P.chunks + c);
  const SeqDesc d = load_desc(P.desc + ch.seq);
  const int r0 = j * d.enum_ns;
  if (r0 >= ch.ns) return;
  const int r1 = min(r0 + d.enum_ns, ch.ns);
  const int L2 = d.len2;
  const int nruns = (L2 + 63) >> 6;
  if (R * w >= nruns) return;  // (banded numbering: this Seq2 has fewer levels than the launch's longest)
  const int qend = 64 * nruns;           // row length
  const int qlast = (L2 + 15) & ~15;     // the candidate loop stops at the 16-value step that holds q = L2 - 1
  const int pc0 = w * R;
  const int rowlen = qend + 64;  // a row of g and its run of padding
  const int *__restrict__ G = fg + ((size_t)ch.fg_row + (size_t)r0 * (nruns + 1)) * 64;  // row of offset r0
  // Register r holds run pc0 + r of p; past the last run it repeats that run (its maxima are
  // dropped below).  f of an offset is the next offset's row of g minus that row's constant.
  // The per-lane loads are buffer loads: the unit's first row is the base, a scalar byte offset
  // walks the rows, and one register per run holds the lane's byte offset within a row -- no
  // 64-bit address arithmetic in vector registers.
  const __amdgpu_buffer_rsrc_t rows = __builtin_amdgcn_make_buffer_rsrc((void *)G, 0, -1, 0x00020000);
  int vb[R];
#pragma unroll
  for (int r = 0; r < R; ++r) vb[r] = 4 * (64 * min(pc0 + r, nruns - 1) + lane);
  int soff = 0;  // byte offset of the current row (at most kEnumMaxNs + 1 rows of 128 + 1 runs)
  int fn[R], acc[R][2];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    fn[r] = __builtin_amdgcn_raw_buffer_load_b32(rows, vb[r], soff + 4 * rowlen, 0);
    acc[r][0] = acc[r][1] = Cand<F32>::lowest;
  }
  int tn = G[rowlen + qend];  // the constant of the next row
  for (int i = r0; i < r1; ++i) {
    int f[R];
#pragma unroll
    for (int r = 0; r < R; ++r) f[r] = Cand<F32>::sub(fn[r], tn);
    const int goff = soff + 4;  // g[p + 1] per lane: the diagonal's left-over pairs
    soff += 4 * rowlen;
    if (i + 1 < r1) {  // next offset's f values arrive while this one is scored
#pragma unroll
      for (int r = 0; r < R; ++r) fn[r] = __builtin_amdgcn_raw_buffer_load_b32(rows, vb[r], soff + 4 * rowlen, 0);
      tn = G[2 * rowlen + qend];
    }
    int gnext[R];
#pragma unroll
    for (int r = 0; r < R; ++r) gnext[r] = __builtin_amdgcn_raw_buffer_load_b32(rows, vb[r], goff, 0);
    wave_full<R, F32>(G, 64 * (pc0 + R), qlast, f, acc);
    {
      // the level's own runs, read as one ascending stream of 16-value steps
      int ga[16], gb[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) ga[k] = G[64 * pc0 + k];
      WaveOwn<R, R - 1, F32>::run(G, pc0, nruns, f, acc, ga, gb);
    }
    if ((lane & 1) == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r][0] = Cand<F32>::max2(acc[r][0], Cand<F32>::add(f[r], gnext[r]));
    }
    G += rowlen;
  }
  // best fixed-point score of this unit and the smallest p attaining it
  int bestv = INT_MIN;
  uint32_t bestp = 0x7fffffffu;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int p = 64 * (pc0 + r) + lane;
    const int a = Cand<F32>::to_int(Cand<F32>::max2(acc[r][0], acc[r][1]));
    if (p <= L2 - 2 && a > bestv) {
      bestv = a;
      bestp = (uint32_t)p;
    }
  }
  unsigned long long key = ((unsigned long long)((uint32_t)bestv ^ 0x80000000u) << 32) | (0xffffffffu - bestp);
  key = wave_max_u64(key);
  if (lane == 0) {
    // a stale read can only be too small, which costs an atomic, never a result
    if (key > P.chunk_best[c]) atomicMax(&P.chunk_best[c], key);  // lets the locate pass skip chunks
    if (key > P.acc[ch.seq].enum_key) atomicMax(&P.acc[ch.seq].enum_key, key);
  }
}

// A workgroup is 1 to 8 independent waves, each scoring one unit: no barrier, no LDS traffic.
// Workgroups are launched in unit order and the SIMD issues its oldest wave first, so units
// finish roughly in the order they were numbered (largest first) and the last waves of a
// launch are short ones.  (Resident waves that draw unit numbers from a counter were
// measured 25 % slower: the issue order starves the youngest waves of each SIMD, which
// then hold their first, largest units until every other wave has run out of work.)
// The launch may ask for LDS it never uses: 41 KB pins three workgroups of 8 waves per CU,
// which leaves wave slots, registers and LDS for the short kernels of the other streams.
// A launch that takes every wave slot uses workgroups of one wave: a workgroup is placed only
// when all its waves find slots on one CU, so with 8-wave workgroups the slots of finished
// waves idle until 8 are free (units of mixed lengths: 64.8 against 55.4 ms on 256 Seq2 of
// 500 to 2000 characters).
template <int R, bool F32>
__global__ __launch_bounds__(512, 8) void psa_enum_wave(const int *__restrict__ fg, WaveParams P) {
  const int lane = threadIdx.x & 63;
  const int u = uni((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
  if (u < P.nunits) wave_unit<R, F32>(fg, P, u, lane);
}

// The same units with the long phase on the matrix cores (float32 arithmetic only).  The result
// tiles and the operands of 8 blocks of positions take registers: 4 waves per SIMD instead of 8,
// which this form does not need (its vector instructions are independent of the matrix core's).
template <bool STAMP>
__global__ __launch_bounds__(512, 4) void psa_enum_wave_mfma(const int *__restrict__ fg, WaveParams P) {
  const int lane = threadIdx.x & 63;
  const int u = uni((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
  if (u < P.nunits) wave_unit_mfma<STAMP>(fg, P, u, lane);
}

// waves per SIMD the locate kernel's register budget is sized for
constexpr int enum_waves_per_simd(int nt, int r) { return nt * r > 4096 ? 4 : 8; }

// maximum over the wave, in every lane: rows of 16 lanes by row shifts (a lane without a source keeps its own
// value), then lane 15 of rows 0 and 2 into rows 1 and 3, lane 31 into rows 2 and 3; lane 63 holds the result
__device__ __forceinline__ int wave_max_i32(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));  // row_shr:1
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));  // row_shr:2
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));  // row_shr:4
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));  // row_shr:8
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));  // row_bcast:15
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));  // row_bcast:31
  return __builtin_amdgcn_readlane(v, 63);
}

// Second pass, two jobs on the chunks worth revisiting (nearly every chunk is skipped after three
// loads):
// * locate: knowing the maximum G and the smallest p* that attains it, find the smallest q and
//   then the smallest offset with f_s[p*] + g_s[q] == G.  Only chunks whose own (maximum, p)
//   equals the Seq2's can contain such a candidate.
// * shortlist: every (offset, p) that has a candidate with f_s[p] + g_s[q] >= G - slack becomes a
//   row of the run's list, with the mask of the 256-wide blocks of q that hold one, to be replayed
//   in float32 (launch_short_replay).  Only chunks whose maximum reaches that window can hold
//   one.  Per offset the block takes the maxima of f and of g (an offset whose two maxima do not
//   add up to the window is done), lists the positions p that could reach the window with the
//   best g behind them, and walks the row of g once per listed p.  A Seq2 with more rows than
//   the batch's quota is given up (its counter stays above that): on a low-entropy input every
//   row can lie inside the window.  The list has room for every Seq2's full quota, so which Seq2
//   are given up does not depend on the order in which blocks append their rows.
template <int NT, int R>
__global__ __launch_bounds__(NT, enum_waves_per_simd(NT, R)) void psa_enum_locate(EnumParams P) {
  using L = EnumLds<NT, R>;
  constexpr int NW = NT / 64;
  constexpr int kNone = -(1 << 29);  // below every f and g; two of them still add up inside int32
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  unsigned long long *red = (unsigned long long *)(smem + L::red);
  int *wmax = (int *)(smem + L::slist);  // [0,16): f
  int *plist_n = wmax + 32;  // [0] positions listed, [1] the Seq2 has been given up
  int *plist = wmax + 36;
  unsigned *pmask = (unsigned *)(plist + kShortPList);
  int *bg = (int *)(pmask + kShortPList);  // max of g over positions [64 i, 64 i + 64)
  int *sg = bg + 128;                      // max of g over positions >= 64 i
  constexpr int NB = R * NW;
  static_assert(NB <= 128, "two values per lane in the suffix scan");
  const unsigned cap = (unsigned)P.short_quota;
  // The next-but-one candidate loop waits on this kernel's stream while the next one fills every
  // SIMD with older waves: at the default priority this kernel's waves are issued last (steps of
  // 4 x Seq2 = 2000 enqueued back to back: 0.176 ms per step, 0.169 with the priority raised).
  __builtin_amdgcn_s_setprio(3);

  for (int c = blockIdx.x; c < P.nchunks; c += gridDim.x) {
    const Chunk ch = load_chunk(P.chunks + c);
    const unsigned long long ek = P.acc[ch.seq].enum_key;
    const unsigned long long ck = P.chunk_best[c];
    const int G = uni((int)((uint32_t)(ek >> 32) ^ 0x80000000u));
    // the score the shortlist window hangs below: the Seq2's enumerated maximum, or for one shard of a
    // long pair the maximum over all shards (never below the shard's own)
    const int Gref = P.has_short_ref ? max(G, P.short_ref) : G;
    int slack = -1;
    if (P.short_rows) slack = uni(P.desc[ch.seq].enum_slack);
    // This kernel is the last reader of chunk_best in a run and leaves it zeroed for the next.
    // A wave that reads the zero instead of ck takes the same branch: ek is never 0, and the
    // zero decodes to the lowest value, outside every window.
    const bool exact = uni((int)(ek == ck));
    bool near = uni((int)(slack >= 0 && (int)((uint32_t)(ck >> 32) ^ 0x80000000u) >= Gref - slack));
    if (!exact && !near) {
      if (tid == 0) P.chunk_best[c] = 0ull;
      continue;
    }
    const SeqDesc d = load_desc(P.desc + ch.seq);
    const int L2 = d.len2;
    int bro[R];
    int win0;
    __syncthreads();
    if (tid == 0) P.chunk_best[c] = 0ull;
    enum_chunk_setup<NT, R>(smem, P, ch, d, bro, win0);
    const int pstar = uni((int)(0xffffffffu - (uint32_t)ek));
    const int T = Gref - slack;
    unsigned *const scount = &P.acc[ch.seq].short_count;
    unsigned long long best = ~0ull;
    for (int s = ch.s0; s < ch.s0 + ch.ns; ++s) {
      scan_diagonal<NT, R>(smem, bro, s, win0);
      const int *Es = ring_slot<NT, R>(smem, s);
      const int *E1 = ring_slot<NT, R>(smem, s - 1);
      const int *E2 = ring_slot<NT, R>(smem, s - 2);
      const int tot2 = E2[L2];
      if (exact) {
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
      if (near) {
        int gq[R];
        int mf = kNone;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int x = r * NT + tid;
          gq[r] = (x >= 1 && x <= L2 - 1) ? E1[x] + tot2 - E2[x + 1] : kNone;
          const int fx = x <= L2 - 2 ? Es[x] - E1[x + 1] : kNone;
          mf = max(mf, fx);
          const int wg = wave_max_i32(gq[r]);
          if (lane == 0) bg[r * NW + w] = wg;
        }
        mf = wave_max_i32(mf);
        if (lane == 0) wmax[w] = mf;
        if (tid == 0) {
          *plist_n = 0;
          // whether the list has overflowed meanwhile is read by one thread: the waves of the block
          // must agree on it, there are barriers behind the decision
          plist_n[1] = __atomic_load_n(scount, __ATOMIC_RELAXED) > cap;
        }
        __syncthreads();
        if (uni(plist_n[1])) near = false;
        if (w == 0) {  // suffix maxima of the runs' maxima: position p can only pair with a q behind it
          int v0 = lane < NB ? bg[lane] : kNone, v1 = 64 + lane < NB ? bg[64 + lane] : kNone;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int t0 = __shfl_down(v0, o), t1 = __shfl_down(v1, o);
            if (lane + o < 64) {
              v0 = max(v0, t0);
              v1 = max(v1, t1);
            }
          }
          v0 = max(v0, __shfl(v1, 0));
          sg[lane] = v0;
          sg[64 + lane] = v1;
        }
        __syncthreads();
        int MF = wmax[0];
        for (int i = 1; i < NW; ++i) MF = max(MF, wmax[i]);
        if (uni((int)(near && MF + sg[0] >= T))) {
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int x = r * NT + tid;
            const int fx = x <= L2 - 2 ? Es[x] - E1[x + 1] : kNone;
            // (against the best g from the start of x's own run of 64 on: a few positions too many)
            if (fx + sg[r * NW + w] >= T) {
              const int k = atomicAdd(plist_n, 1);
              if (k < kShortPList) plist[k] = x;
            }
          }
          for (int k = tid; k < kShortPList; k += NT) pmask[k] = 0u;
          __syncthreads();
          int np = *plist_n;
          if (np > kShortPList) {  // more positions than the list holds: give this Seq2 up
            if (tid == 0) atomicMax(scount, cap + 1u);
            np = 0;
          }
          for (int k = 0; k < np; ++k) {
            const int p = plist[k];
            const int fp = Es[p] - E1[p + 1];
            // the 64 positions a wave holds in register r lie in one 256-wide block of q: its bit is set if any
            // lane has a candidate (a compare into a scalar mask -- no exchange between lanes)
            unsigned mask = 0u;
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const int q = r * NT + tid;
              if (__ballot(q > p && fp + gq[r] >= T)) mask |= 1u << ((r * NT + 64 * w) >> 8);
            }
            if (lane == 0 && mask) atomicOr(&pmask[k], mask);
          }
          __syncthreads();
          for (int k = tid; k < np; k += NT) {
            const unsigned mask = pmask[k];
            if (!mask) continue;  // the best g lies at or before p
            if (atomicAdd(scount, 1u) >= cap) continue;
            const unsigned idx = atomicAdd(P.short_total, 1u);
            if (idx < (unsigned)P.short_rows_cap) {
              ShortRow row;
              row.seq = ch.seq;
              row.s = s;
              row.p = (uint32_t)plist[k];
              row.qmask = mask;
              row.best = 0ull;
              P.short_rows[idx] = row;
            } else {
              atomicMax(scount, cap + 1u);  // (cannot happen: the list holds every Seq2's quota)
            }
          }
          // (the next offset's scan has barriers between these reads of the list and its reset)
        }
      }
      if (!near && !exact) break;
    }
    if (exact) {
      best = wave_min_u64(best);
      if (lane == 0) red[w] = best;
      __syncthreads();
      if (tid == 0) {
        for (int i = 1; i < NW; ++i) best = red[i] < best ? red[i] : best;
        if (best != ~0ull) atomicMin(&P.acc[ch.seq].enum_loc, best);
      }
    }
  }
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the (kernel, device) pair: it is set for
// every kernel variant on the device that is current, once per context (set_enum_kernel_attributes,
// called by psa_init), not remembered in a process-wide flag.
template <int NT, int R>
static hipError_t set_lds_attr() {
  using L = EnumLds<NT, R>;  // (the pre-pass needs less than the default limit)
  return hipFuncSetAttribute((const void *)psa_enum_locate<NT, R>, hipFuncAttributeMaxDynamicSharedMemorySize, L::total);
}

hipError_t set_enum_kernel_attributes() {
  hipError_t e;
  if ((e = set_lds_attr<256, 1>()) != hipSuccess) return e;
  if ((e = set_lds_attr<256, 2>()) != hipSuccess) return e;
  if ((e = set_lds_attr<512, 2>()) != hipSuccess) return e;
  if ((e = set_lds_attr<512, 4>()) != hipSuccess) return e;
  if ((e = set_lds_attr<1024, 4>()) != hipSuccess) return e;
  return set_lds_attr<1024, 8>();
}

template <int NT, int R>
static hipError_t launch_prep_t(const EnumParams &P, bool f32, hipStream_t st, hipEvent_t ev_start,
                                hipEvent_t ev_stop) {
  hipError_t e;
  if (f32)
    hipExtLaunchKernelGGL((psa_enum_prep<NT, R, true>), dim3(P.nchunks), dim3(NT), (PrepBlockLds<NT, R>::total), st,
                          ev_start, ev_stop, 0, P);
  else
    hipExtLaunchKernelGGL((psa_enum_prep<NT, R, false>), dim3(P.nchunks), dim3(NT), (PrepBlockLds<NT, R>::total), st,
                          ev_start, ev_stop, 0, P);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return debug_sync("psa_enum_prep");
}

template <int NT, int R>
static hipError_t launch_locate_t(const EnumParams &P, int grid, hipStream_t st) {
  hipError_t e;
  hipLaunchKernelGGL((psa_enum_locate<NT, R>), dim3(grid), dim3(NT), (EnumLds<NT, R>::total), st, P);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return debug_sync("psa_enum_locate");
}

hipError_t launch_enum_wave(const int *fg, const WaveParams &P, bool f32, bool use_mfma, int waves_per_group, int lds_bytes,
                            hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (waves_per_group < 1 || waves_per_group > 8) return hipErrorInvalidValue;
  const dim3 grid((P.nunits + waves_per_group - 1) / waves_per_group), block(64 * waves_per_group);
  // the events ride on the kernel's own dispatch packet: no marker packets before or after it
  if (f32 && use_mfma && P.clock_probe)
    hipExtLaunchKernelGGL(psa_enum_wave_mfma<true>, grid, block, lds_bytes, st, ev_start, ev_stop, 0, fg, P);
  else if (f32 && use_mfma)
    hipExtLaunchKernelGGL(psa_enum_wave_mfma<false>, grid, block, lds_bytes, st, ev_start, ev_stop, 0, fg, P);
  else if (f32)
    hipExtLaunchKernelGGL((psa_enum_wave<kEnumWaveRuns, true>), grid, block, lds_bytes, st,
                          ev_start, ev_stop, 0, fg, P);
  else
    hipExtLaunchKernelGGL((psa_enum_wave<kEnumWaveRuns, false>), grid, block, lds_bytes, st,
                          ev_start, ev_stop, 0, fg, P);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return debug_sync("psa_enum_wave");
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

template <int R>
static hipError_t launch_prep_wave_t(const EnumParams &P, bool f32, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (f32)
    hipExtLaunchKernelGGL((psa_enum_prep_wave<R, true>), dim3(P.nchunks), dim3(64), (PrepWaveLds<R>::total), st, ev_start,
                          ev_stop, 0, P);
  else
    hipExtLaunchKernelGGL((psa_enum_prep_wave<R, false>), dim3(P.nchunks), dim3(64), (PrepWaveLds<R>::total), st, ev_start,
                          ev_stop, 0, P);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return debug_sync("psa_enum_prep_wave");
}

hipError_t launch_enum_prep(int cls, const EnumParams &P, bool f32, hipStream_t st, hipEvent_t ev_start,
                            hipEvent_t ev_stop) {
  if (P.nchunks <= 0) return hipSuccess;
  // Seq2 of up to 2047 characters (32 runs of 64 positions) in a large batch: one wave per chunk, rows held in
  // registers; small batches and the longer Seq2 use the block-wide form
  const char *const force = getenv("PSA_PREP_FORM");  // tuning aid: "block" or "wave" for every batch
  const bool wave_form = force ? force[0] == 'w' : P.prep_wave_form != 0;
  if (wave_form) {
    switch (cls) {
      case 0: return launch_prep_wave_t<4>(P, f32, st, ev_start, ev_stop);
      case 1: return launch_prep_wave_t<8>(P, f32, st, ev_start, ev_stop);
      case 2: return launch_prep_wave_t<16>(P, f32, st, ev_start, ev_stop);
      case 3: return launch_prep_wave_t<32>(P, f32, st, ev_start, ev_stop);
      default: break;
    }
  }
  switch (cls) {
    case 0: return launch_prep_t<256, 1>(P, f32, st, ev_start, ev_stop);
    case 1: return launch_prep_t<256, 2>(P, f32, st, ev_start, ev_stop);
    case 2: return launch_prep_t<512, 2>(P, f32, st, ev_start, ev_stop);
    case 3: return launch_prep_t<512, 4>(P, f32, st, ev_start, ev_stop);
    case 4: return launch_prep_t<1024, 4>(P, f32, st, ev_start, ev_stop);
    case 5: return launch_prep_t<1024, 8>(P, f32, st, ev_start, ev_stop);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_enum_locate(int cls, const EnumParams &P, int grid, hipStream_t st) {
  if (P.nchunks <= 0) return hipSuccess;
  switch (cls) {
    case 0: return launch_locate_t<256, 1>(P, grid, st);
    case 1: return launch_locate_t<256, 2>(P, grid, st);
    case 2: return launch_locate_t<512, 2>(P, grid, st);
    case 3: return launch_locate_t<512, 4>(P, grid, st);
    case 4: return launch_locate_t<1024, 4>(P, grid, st);
    case 5: return launch_locate_t<1024, 8>(P, grid, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace psa
