// gfx950 kernels of libpsa_hip.
//
// Scored quantity (P/cudaFunctions.cu:205-223, S/main.c:53-61): for offset s and
// deleted 0-based positions p < q of Seq2 (B, length L2) against Seq1 (A),
//   score(s,p,q) = sum_{j<p} T[A[s+j]][B[j]] + sum_{p<j<q} T[A[s+j-1]][B[j]]
//                + sum_{j>q} T[A[s+j-2]][B[j]]
// added left to right in float32.  All three sums run along anti-diagonals
// delta = s, s-1, s-2 of the L1 x L2 comparison grid.
//
// Two families of kernels:
//  * enumeration (psa_enum_main / psa_enum_locate): with E^delta[x] the exclusive
//    prefix sum of the fixed-point table entries along diagonal delta,
//      score = f_s[p] + g_s[q],  f_s[p] = E^s[p] - E^{s-1}[p+1],
//                                g_s[q] = E^{s-1}[q] + E^{s-2}[L2] - E^{s-2}[q+1],
//    so every (s,p,q) costs one integer add and one max.  A workgroup owns a run
//    of offsets; each lane keeps f for R values of p in registers with one
//    running maximum each, g is broadcast from LDS.
//  * float32 resolve (psa_fwd / psa_collect / psa_first_p / psa_first_q): the
//    reference's rounded sum is monotone in its starting value, so a three-state
//    max-plus recurrence over Seq2 positions gives the exact float32 maximum per
//    offset, and two further passes find the first maximiser in (n,k,offset) order.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "psa_kernels.h"

namespace psa {

hipError_t debug_sync(const char *what);

__device__ __forceinline__ uint32_t f2key(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = __shfl_xor(lo, m);
  hi = __shfl_xor(hi, m);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    unsigned long long o = shfl_xor_u64(v, m);
    v = o > v ? o : v;
  }
  return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    unsigned long long o = shfl_xor_u64(v, m);
    v = o < v ? o : v;
  }
  return v;
}

// Values that are the same in every lane are moved to scalar registers before they
// steer a branch around a barrier: hipcc then emits a scalar branch instead of
// structurising the region as divergent (which can let part of a wave run ahead
// to the next barrier).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ Chunk load_chunk(const Chunk *p) {
  Chunk c;
  c.seq = uni(p->seq);
  c.s0 = uni(p->s0);
  c.ns = uni(p->ns);
  c.pad = 0;
  return c;
}
__device__ __forceinline__ SeqDesc load_desc(const SeqDesc *p) {
  SeqDesc d;
  d.fs_base = ((int64_t)uni((int)(p->fs_base >> 32)) << 32) | (uint32_t)uni((int)(uint32_t)p->fs_base);
  d.off2 = uni(p->off2);
  d.len2 = uni(p->len2);
  d.s_begin = uni(p->s_begin);
  d.s_end = uni(p->s_end);
  d.frac_bits = uni(p->frac_bits);
  d.pad = 0;
  return d;
}

// ---------------------------------------------------------------------------------
// per-batch initialisation of the accumulators and counters
// ---------------------------------------------------------------------------------
__global__ void psa_init_acc(SeqAcc *acc, int n, unsigned *counters, int ncounters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    SeqAcc a;
    a.enum_key = 0ull;
    a.enum_loc = ~0ull;
    a.res_loc = ~0ull;
    a.res_key = 0u;
    a.res_p = INT_MAX;
    acc[i] = a;
  }
  if (i < ncounters) counters[i] = 0u;
}

// ---------------------------------------------------------------------------------
// enumeration
// ---------------------------------------------------------------------------------
template <int NT, int R>
struct EnumLds {
  static constexpr int NP = NT * R;  // positions covered (needs NP >= L2 + 1)
  static constexpr int NW = NT / 64;
  // byte offsets inside dynamic LDS
  static constexpr int ring = 0;                    // 3 * NP int32: E^delta for delta mod 3
  static constexpr int g = ring + 3 * NP * 4;       // NP int32
  static constexpr int itab = g + NP * 4;           // kDevTable int32
  static constexpr int wtot = itab + kDevTable * 4; // 16 int32
  static constexpr int red = wtot + 16 * 4;         // 16 u64
  static constexpr int misc = red + 16 * 8;         // 4 int32
  static constexpr int awin = misc + 16;            // kEnumMaxNs + NP + 2 bytes, rounded up
  static constexpr int total = awin + ((kEnumMaxNs + NP + 2 + 15) / 16) * 16;
};

template <int NT, int R>
struct EnumState {
  int bro[R];  // (Seq2 code << 5) of the R consecutive positions tid*R .. tid*R+R-1
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

// One 64-wide run of q against the first K registers, all lanes valid.
template <int R, int K>
__device__ __forceinline__ void enum_full(const int4 *__restrict__ g4, const int (&f)[R], int (&acc)[R]) {
  int4 cur = g4[0];
#pragma unroll 2
  for (int u = 0; u < 16; ++u) {
    int4 nxt = g4[u < 15 ? u + 1 : 15];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r < K) {
        int t0 = f[r] + cur.x, t1 = f[r] + cur.y, t2 = f[r] + cur.z, t3 = f[r] + cur.w;
        acc[r] = max(max(acc[r], t0), t1);
        acc[r] = max(max(acc[r], t2), t3);
      }
    }
    cur = nxt;
  }
}

// The run of q that overlaps register K's own p range: lane l holds p = q0 + l,
// so q0 + i pairs only with lanes l < i.
template <int R, int K>
__device__ __forceinline__ void enum_diag(const int4 *__restrict__ g4, const int (&f)[R], int (&acc)[R], int lane) {
  const int fk = f[K];
  int a = acc[K];
#pragma unroll 4
  for (int u = 0; u < 16; ++u) {
    int4 cur = g4[u];
    int i = u * 4;
    int t0 = fk + cur.x, t1 = fk + cur.y, t2 = fk + cur.z, t3 = fk + cur.w;
    if (lane < i) a = max(a, t0);
    if (lane < i + 1) a = max(a, t1);
    if (lane < i + 2) a = max(a, t2);
    if (lane < i + 3) a = max(a, t3);
  }
  acc[K] = a;
}

template <int R, int K>
struct FullDispatch {
  __device__ static __forceinline__ void run(int nf, const int4 *g4, const int (&f)[R], int (&acc)[R]) {
    if (nf == K)
      enum_full<R, K>(g4, f, acc);
    else
      FullDispatch<R, K - 1>::run(nf, g4, f, acc);
  }
};
template <int R>
struct FullDispatch<R, 0> {
  __device__ static __forceinline__ void run(int, const int4 *, const int (&)[R], int (&)[R]) {}
};
template <int R, int K>
struct DiagDispatch {
  __device__ static __forceinline__ void run(int rd, const int4 *g4, const int (&f)[R], int (&acc)[R], int lane) {
    if (rd == K)
      enum_diag<R, K>(g4, f, acc, lane);
    else
      DiagDispatch<R, K - 1>::run(rd, g4, f, acc, lane);
  }
};
template <int R>
struct DiagDispatch<R, -1> {
  __device__ static __forceinline__ void run(int, const int4 *, const int (&)[R], int (&)[R], int) {}
};

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

template <int NT, int R>
__global__ __launch_bounds__(NT) void psa_enum_main(EnumParams P) {
  using L = EnumLds<NT, R>;
  constexpr int NW = NT / 64;
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
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
    const int QC = (L2 - 1) / 64 + 1;

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
      for (int qc = 0; qc < QC; ++qc) {
        const int4 *g4 = (const int4 *)(gL + qc * 64);
        int nf = (qc + NW - 1 - w) / NW;  // registers whose whole p run lies below this q run
        if (nf > R) nf = R;
        FullDispatch<R, R>::run(nf, g4, f, acc);
        const int dq = qc - w;
        if (dq >= 0 && dq % NW == 0 && dq / NW < R) DiagDispatch<R, R - 1>::run(dq / NW, g4, f, acc, lane);
      }
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
      atomicMax(&P.acc[ch.seq].enum_key, key);
      misc[0] = (int)atomicAdd(P.counter, 1u);  // next chunk
    }
    __syncthreads();
    c = uni(misc[0]);
  }
}

// Second pass: knowing the maximum G and the smallest p* that attains it, find the
// smallest q and then the smallest offset with f_s[p*] + g_s[q] == G.
template <int NT, int R>
__global__ __launch_bounds__(NT) void psa_enum_locate(EnumParams P) {
  using L = EnumLds<NT, R>;
  constexpr int NW = NT / 64;
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  unsigned long long *red = (unsigned long long *)(smem + L::red);

  for (int c = blockIdx.x; c < P.nchunks; c += gridDim.x) {
    const Chunk ch = load_chunk(P.chunks + c);
    const SeqDesc d = load_desc(P.desc + ch.seq);
    const int L2 = d.len2;
    int bro[R];
    int win0;
    __syncthreads();
    enum_chunk_setup<NT, R>(smem, P, ch, d, bro, win0);
    const unsigned long long ek = P.acc[ch.seq].enum_key;
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

// ---------------------------------------------------------------------------------
// float32 resolve, pass 1: per-offset maximum over all (n,k); lanes = offsets.
// State before Seq2 position j: a (no deletion yet), b (one), c (two).
//   a' = a + T[A[s+j]][B[j]]
//   b' = max(a, b + T[A[s+j-1]][B[j]])      delete j, or keep it after one deletion
//   c' = max(b, c + T[A[s+j-2]][B[j]])
// Rounding is monotone, so taking the max before continuing the chain is exact.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void psa_fwd(ResolveParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  FwdBlock fb;
  fb.seq = uni(P.fwd_blocks[blockIdx.x].seq);
  fb.s0 = uni(P.fwd_blocks[blockIdx.x].s0);
  const SeqDesc d = load_desc(P.desc + fb.seq);
  const int L2 = d.len2;
  float *T = (float *)smem;                                  // kDevTable floats
  unsigned char *bc = smem + kDevTable * 4;                  // L2 codes, padded to 16
  unsigned char *aw = bc + ((L2 + 15) / 16) * 16;            // 256 + L2 + 2 codes
  uint32_t *red = (uint32_t *)(aw + ((258 + L2 + 15) / 16) * 16);
  for (int i = tid; i < kDevTable; i += 256) T[i] = P.table_t[i];
  for (int i = tid; i < L2; i += 256) bc[i] = P.seq2_pool[d.off2 + i];
  const int win0 = fb.s0 - 2;
  for (int i = tid; i < 258 + L2; i += 256) {
    int64_t idx = (int64_t)win0 + i;
    aw[i] = (idx >= 0 && idx < P.len1) ? P.seq1[idx] : (unsigned char)kPadCode;
  }
  __syncthreads();
  const int s = fb.s0 + tid;
  const float NEG = -INFINITY;
  float a = 0.0f, b = NEG, c = NEG;
  int a2 = aw[tid], a1 = aw[tid + 1];
  const unsigned char *ap = aw + tid + 2;
  int j = 0;
#pragma unroll 4
  for (; j <= L2 - 3; ++j) {
    const int a0 = ap[j];
    const float *row = T + (int)bc[j] * kRowStride;
    const float x0 = row[a0], x1 = row[a1], x2 = row[a2];
    const float nc = fmaxf(b, c + x2);
    const float nb = fmaxf(a, b + x1);
    a = a + x0;
    b = nb;
    c = nc;
    a2 = a1;
    a1 = a0;
  }
  if (j == L2 - 2) {  // no further undeleted step: two deletions must still fit
    const int a0 = ap[j];
    const float *row = T + (int)bc[j] * kRowStride;
    const float x1 = row[a1], x2 = row[a2];
    const float nc = fmaxf(b, c + x2);
    b = fmaxf(a, b + x1);
    c = nc;
    a2 = a1;
    a1 = a0;
    ++j;
  }
  {  // j == L2 - 1: the last position is either the second deletion or consumed
    const float *row = T + (int)bc[j] * kRowStride;
    c = fmaxf(b, c + row[a2]);
  }
  uint32_t key = 0u;
  if (s < d.s_end) {
    P.fs[d.fs_base + (s - d.s_begin)] = c;
    key = f2key(c);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    uint32_t o = __shfl_xor(key, m);
    key = o > key ? o : key;
  }
  if (lane == 0) red[w] = key;
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < 4; ++i) key = red[i] > key ? red[i] : key;
    atomicMax(&P.acc[fb.seq].res_key, key);
  }
}

// pass 2: list the offsets whose maximum equals the Seq2's maximum
__global__ __launch_bounds__(256) void psa_collect(ResolveParams P) {
  const FwdBlock fb = P.fwd_blocks[blockIdx.x];
  const SeqDesc d = P.desc[fb.seq];
  const int s = fb.s0 + threadIdx.x;
  if (s < d.s_end) {
    const float v = P.fs[d.fs_base + (s - d.s_begin)];
    if (f2key(v) == P.acc[fb.seq].res_key) {
      unsigned idx = atomicAdd(P.nactive, 1u);
      ActiveOffset ao;
      ao.seq = fb.seq;
      ao.s = s;
      P.active[idx] = ao;
    }
  }
}

// Table entries seen by Seq2 position j at offset s after 0, 1, 2 deletions.
__device__ __forceinline__ void fill_diagonals(float4 *D, const float *T, const ResolveParams &P, const SeqDesc &d,
                                               int s) {
  const int L2 = d.len2;
  for (int j = threadIdx.x; j < L2; j += blockDim.x) {
    const int b = P.seq2_pool[d.off2 + j];
    const float *row = T + b * kRowStride;
    int64_t i0 = (int64_t)s + j;
    int c0 = (i0 < P.len1) ? P.seq1[i0] : kPadCode;
    int c1 = (i0 - 1 >= 0 && i0 - 1 < P.len1) ? P.seq1[i0 - 1] : kPadCode;
    int c2 = (i0 - 2 >= 0 && i0 - 2 < P.len1) ? P.seq1[i0 - 2] : kPadCode;
    D[j] = make_float4(row[c0], row[c1], row[c2], 0.0f);
  }
}

// pass 3: for an offset that attains the maximum, lane p computes
// max_q score(s,p,q) exactly (same monotone recurrence with the first deletion
// pinned to p); the smallest p that attains the maximum wins.
__global__ __launch_bounds__(256) void psa_first_p(ResolveParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int skip_flag[1];
  float *T = (float *)smem;
  float4 *D = (float4 *)(smem + kDevTable * 4);
  const int tid = threadIdx.x;
  for (int i = tid; i < kDevTable; i += 256) T[i] = P.table_t[i];
  const unsigned nact = *P.nactive;
  const long long nitems = (long long)nact * P.pchunks;
  for (long long item = blockIdx.x; item < nitems; item += gridDim.x) {
    ActiveOffset ao;
    ao.seq = uni(P.active[item / P.pchunks].seq);
    ao.s = uni(P.active[item / P.pchunks].s);
    const int pc = (int)(item % P.pchunks);
    const SeqDesc d = load_desc(P.desc + ao.seq);
    const int L2 = d.len2;
    const int p0 = pc * 256;
    if (p0 > L2 - 2) continue;
    __syncthreads();
    if (tid == 0) skip_flag[0] = p0 > P.acc[ao.seq].res_p;  // a smaller p is already known
    __syncthreads();
    if (uni(skip_flag[0])) continue;  // one decision for the whole block
    fill_diagonals(D, T, P, d, ao.s);
    __syncthreads();
    const float fstar = key2f(P.acc[ao.seq].res_key);
    const int p = p0 + tid;
    float v = 0.0f, c = -INFINITY;
    int j = 0;
    for (; j < p0 && j < L2; ++j) v += D[j].x;  // every lane of the block is still undeleted
    for (; j < L2; ++j) {
      const float4 x = D[j];
      const float nc = fmaxf(v, c + x.z);
      const float nv = v + (j < p ? x.x : x.y);
      c = j > p ? nc : c;
      v = j == p ? v : nv;
    }
    int cand = (p <= L2 - 2 && c == fstar) ? p : INT_MAX;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) cand = min(cand, __shfl_xor(cand, m));
    if ((tid & 63) == 0 && cand != INT_MAX) atomicMin(&P.acc[ao.seq].res_p, cand);
  }
}

// pass 4: with p* fixed, lane q evaluates score(s,p*,q) exactly as the reference
// adds it; the smallest q, then the smallest offset, wins.
__global__ __launch_bounds__(256) void psa_first_q(ResolveParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int skip_flag[1];
  float *T = (float *)smem;
  float4 *D = (float4 *)(smem + kDevTable * 4);
  const int tid = threadIdx.x;
  for (int i = tid; i < kDevTable; i += 256) T[i] = P.table_t[i];
  const unsigned nact = *P.nactive;
  const long long nitems = (long long)nact * P.pchunks;
  for (long long item = blockIdx.x; item < nitems; item += gridDim.x) {
    ActiveOffset ao;
    ao.seq = uni(P.active[item / P.pchunks].seq);
    ao.s = uni(P.active[item / P.pchunks].s);
    const int qc = (int)(item % P.pchunks);
    const SeqDesc d = load_desc(P.desc + ao.seq);
    const int L2 = d.len2;
    const int pstar = uni(P.acc[ao.seq].res_p);
    const int q0 = qc * 256;
    if (pstar == INT_MAX || q0 > L2 - 1 || q0 + 255 <= pstar) continue;
    __syncthreads();
    if (tid == 0) {
      const unsigned long long known = P.acc[ao.seq].res_loc;
      skip_flag[0] = known != ~0ull && (int)(known >> 32) < q0;  // a smaller q is already known
    }
    __syncthreads();
    if (uni(skip_flag[0])) continue;  // one decision for the whole block
    fill_diagonals(D, T, P, d, ao.s);
    __syncthreads();
    const float fstar = key2f(P.acc[ao.seq].res_key);
    const int q = q0 + tid;
    float v = 0.0f, c = 0.0f;
    int j = 0;
    for (; j < pstar; ++j) v += D[j].x;
    j = pstar + 1;
    for (; j < L2; ++j) {
      const float4 x = D[j];
      const float nv = v + x.y;
      const float nc = c + x.z;
      c = j == q ? v : (j > q ? nc : c);
      v = j < q ? nv : v;
    }
    unsigned long long cand = ~0ull;
    if (q > pstar && q <= L2 - 1 && c == fstar) cand = ((unsigned long long)(uint32_t)q << 32) | (uint32_t)ao.s;
    cand = wave_min_u64(cand);
    if ((tid & 63) == 0 && cand != ~0ull) atomicMin(&P.acc[ao.seq].res_loc, cand);
  }
}

// ---------------------------------------------------------------------------------
// result records: decode the accumulators and count residue classes of the winners
// ---------------------------------------------------------------------------------
__device__ void count_classes(const FinalParams &P, const SeqDesc &d, int s, int p, int q, int *lds_counts,
                              int32_t *out) {
  const int tid = threadIdx.x;
  if (tid < 4) lds_counts[tid] = 0;
  __syncthreads();
  int local[4] = {0, 0, 0, 0};
  const int m = d.len2 - 2;
  for (int i = tid; i < m; i += blockDim.x) {
    const int j = i < p ? i : (i < q - 1 ? i + 1 : i + 2);
    const int a = P.seq1[(int64_t)s + i], b = P.seq2_pool[d.off2 + j];
    const int cls = P.class_t[b * kRowStride + a];
    local[0] += cls == 0;
    local[1] += cls == 1;
    local[2] += cls == 2;
    local[3] += cls == 3;
  }
  for (int k = 0; k < 4; ++k)
    if (local[k]) atomicAdd(&lds_counts[k], local[k]);
  __syncthreads();
  if (tid < 4) out[tid] = lds_counts[tid];
  __syncthreads();
}

__global__ __launch_bounds__(256) void psa_finalize(FinalParams P) {
  __shared__ int lds_counts[4];
  const int i = blockIdx.x;
  const SeqDesc d = P.desc[i];
  const SeqAcc a = P.acc[i];
  psa_result *r = P.out + i;
  const int tid = threadIdx.x;
  const bool res_ok = uni((P.stages & PSA_STAGE_RESOLVE) && a.res_loc != ~0ull && a.res_p != INT_MAX);
  const bool enum_ok = uni((P.stages & PSA_STAGE_ENUMERATE) && d.frac_bits >= 0 && a.enum_loc != ~0ull);
  if (res_ok) {
    const int p = a.res_p, q = (int)(a.res_loc >> 32), s = (int)(uint32_t)a.res_loc;
    count_classes(P, d, s, p, q, lds_counts, r->counts);
    if (tid == 0) {
      r->n = p + 1;
      r->k = q + 1;
      r->offset = s;
      const float sc = key2f(a.res_key);
      r->score = sc;
      r->score_bits = __float_as_uint(sc);
      const long long L = d.len2, n = p + 1;
      r->mutant_num = (int32_t)((n - 1) * L - n * (n - 1) / 2 + (q - p) - 1);
    }
  } else if (tid == 0) {
    r->n = r->k = r->offset = r->mutant_num = 0;
    r->score = 0.0f;
    r->score_bits = 0u;
    r->counts[0] = r->counts[1] = r->counts[2] = r->counts[3] = 0;
  }
  if (enum_ok) {
    const int p = (int)(0xffffffffu - (uint32_t)a.enum_key);
    const int q = (int)(a.enum_loc >> 32), s = (int)(uint32_t)a.enum_loc;
    count_classes(P, d, s, p, q, lds_counts, r->enum_counts);
    if (tid == 0) {
      r->enum_n = p + 1;
      r->enum_k = q + 1;
      r->enum_offset = s;
      r->enum_score = (int)((uint32_t)(a.enum_key >> 32) ^ 0x80000000u);
    }
  } else if (tid == 0) {
    r->enum_n = r->enum_k = r->enum_offset = r->enum_score = 0;
    r->enum_counts[0] = r->enum_counts[1] = r->enum_counts[2] = r->enum_counts[3] = 0;
  }
}

// ---------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------
// With PSA_DEBUG_SYNC set in the environment every launch is followed by a device
// synchronisation and a line on stderr naming the kernel that just finished.
hipError_t debug_sync(const char *what) {
  static const bool on = getenv("PSA_DEBUG_SYNC") != nullptr;
  if (!on) return hipSuccess;
  hipError_t e = hipDeviceSynchronize();
  fprintf(stderr, "[psa] %s: %s\n", what, hipGetErrorString(e));
  fflush(stderr);
  return e;
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
  static const int np[kEnumClasses] = {256 * 1, 256 * 2, 256 * 4, 256 * 8, 256 * 16, 512 * 16};
  return np[c];
}

int enum_class_lds_bytes(int c) {
  switch (c) {
    case 0: return EnumLds<256, 1>::total;
    case 1: return EnumLds<256, 2>::total;
    case 2: return EnumLds<256, 4>::total;
    case 3: return EnumLds<256, 8>::total;
    case 4: return EnumLds<256, 16>::total;
    default: return EnumLds<512, 16>::total;
  }
}

int enum_class_threads(int c) { return c == 5 ? 512 : 256; }

hipError_t launch_enum(int cls, const EnumParams &P, int grid_main, int grid_locate, hipStream_t st,
                       hipEvent_t ev_begin, hipEvent_t ev_end) {
  switch (cls) {
    case 0: return launch_enum_t<256, 1>(P, grid_main, grid_locate, st, ev_begin, ev_end);
    case 1: return launch_enum_t<256, 2>(P, grid_main, grid_locate, st, ev_begin, ev_end);
    case 2: return launch_enum_t<256, 4>(P, grid_main, grid_locate, st, ev_begin, ev_end);
    case 3: return launch_enum_t<256, 8>(P, grid_main, grid_locate, st, ev_begin, ev_end);
    case 4: return launch_enum_t<256, 16>(P, grid_main, grid_locate, st, ev_begin, ev_end);
    case 5: return launch_enum_t<512, 16>(P, grid_main, grid_locate, st, ev_begin, ev_end);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_init(SeqAcc *acc, int n, unsigned *counters, int ncounters, hipStream_t st) {
  int m = n > ncounters ? n : ncounters;
  hipLaunchKernelGGL(psa_init_acc, dim3((m + 255) / 256), dim3(256), 0, st, acc, n, counters, ncounters);
  return hipGetLastError();
}

static size_t fwd_lds(int max_len2) {
  return (size_t)kDevTable * 4 + ((max_len2 + 15) / 16) * 16 + ((258 + max_len2 + 15) / 16) * 16 + 64;
}
static size_t tie_lds(int max_len2) { return (size_t)kDevTable * 4 + (size_t)max_len2 * 16; }

hipError_t launch_resolve(const ResolveParams &P, int nfwd_blocks, int max_len2, int tie_grid, hipStream_t st) {
  static size_t attr_bytes = 0;
  hipError_t e;
  const size_t tb = tie_lds(max_len2);
  if (tb > attr_bytes) {
    e = hipFuncSetAttribute((const void *)psa_first_p, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tb);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void *)psa_first_q, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tb);
    if (e != hipSuccess) return e;
    attr_bytes = tb;
  }
  hipLaunchKernelGGL(psa_fwd, dim3(nfwd_blocks), dim3(256), fwd_lds(max_len2), st, P);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = debug_sync("psa_fwd")) != hipSuccess) return e;
  hipLaunchKernelGGL(psa_collect, dim3(nfwd_blocks), dim3(256), 0, st, P);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = debug_sync("psa_collect")) != hipSuccess) return e;
  hipLaunchKernelGGL(psa_first_p, dim3(tie_grid), dim3(256), tb, st, P);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = debug_sync("psa_first_p")) != hipSuccess) return e;
  hipLaunchKernelGGL(psa_first_q, dim3(tie_grid), dim3(256), tb, st, P);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return debug_sync("psa_first_q");
}

hipError_t launch_finalize(const FinalParams &P, int n, hipStream_t st) {
  hipLaunchKernelGGL(psa_finalize, dim3(n), dim3(256), 0, st, P);
  return hipGetLastError();
}

}  // namespace psa
