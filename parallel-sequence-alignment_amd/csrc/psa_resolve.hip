// gfx950 float32-exact resolve kernels of libpsa_hip, plus accumulator set-up and the
// result records.
//
// The reference adds the table entries of a candidate left to right in float32
// (P/cudaFunctions.cu:210-220, S/main.c:53-61).  Rounding to nearest is monotone
// (x <= y implies fl(x+d) <= fl(y+d)), so the maximum over all (n,k) for one offset
// follows from a three-state max-plus recurrence over Seq2 positions (psa_fwd), and
// the first maximiser in (n, k, offset) order -- the reference's strict '>' rule
// (S/main.c:74, P/cudaFunctions.cu:241, P/cFunctions.c:129,146) -- from two further
// passes over the offsets that attain the maximum (psa_first_p, psa_first_q).
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>

#include <hip/hip_ext.h>

#include "psa_device.h"

namespace psa {

// Built with -DPSA_TRACE the chain kernels print, per workgroup, the shader cycles between the
// marks PSA_TS() in their code (s_memtime) -- where the microseconds of a latency-bound kernel go.
#ifdef PSA_TRACE
#define PSA_TS_DECL() unsigned long long ts_[12]; int nts_ = 0;
#define PSA_TS() ts_[nts_++] = __builtin_amdgcn_s_memtime();
#else
#define PSA_TS_DECL()
#define PSA_TS()
#endif

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
    a.short_best = 0ull;
    a.short_s = ~0u;
    a.short_count = 0u;
    acc[i] = a;
  }
  if (i < ncounters) counters[i] = 0u;
}

// ---------------------------------------------------------------------------------
// float32 resolve, pass 1: per-offset maximum over all (n,k); lanes = offsets.
// State of offset s before Seq2 position j: a (no deletion yet), b (one), c (two).
//   a' = a + x0,  b' = max(a, b + x1),  c' = max(b, c + x2),   x_d = T[A[s+j-d]][B[j]]
// (delete j, or keep it after d deletions).  Rounding is monotone, so taking the max
// before continuing the chain is exact.
//
// x1 of offset s is x0 of offset s - 1 and x2 is x0 of offset s - 2 at the same step, so a lane
// keeps the three states of three CONSECUTIVE offsets that meet the same table entry: lane l
// of a wave reads x0 of offset u = base + l and holds a of u, b of u + 1 and c of u + 2.  Then
//   a_u += x0,  b_{u+1} = max(a_{u+1}, b_{u+1} + x0),  c_{u+2} = max(b_{u+2}, c_{u+2} + x0),
// where a_{u+1} and b_{u+2} are the values the next lane holds: they arrive as the shifted
// operand of the max itself (DPP wave_shl:1), b and c advance with one packed add.  A step is
// one LDS table read, its address add and four vector instructions.  The last two lanes of a
// wave have no complete right neighbours, so a wave yields 62 offsets and a block
// kFwdOffsets = 248.
// The kernel runs one wave per SIMD and is bound by how fast that wave can issue, i.e.
// by instructions per step: the Seq1 codes and the Seq2 row offsets of a group of four
// steps are fetched two groups ahead, the table entries one group ahead, so no LDS
// latency sits on the dependent chain.
// (This file is compiled with -fno-honor-nans: no NaN can arise here -- finite table entries,
// and -inf only ever meets finite values -- and without the flag every max would first
// canonicalise its operands, one more instruction on the chain each.)
//
// LDS: table | row byte offsets of Seq2 (128 * code per position) | Seq1 window as byte
// offsets (4 * code)
// ---------------------------------------------------------------------------------
__host__ __device__ inline int fwd_rows(int len2) { return ((len2 + 3) / 4) * 4 + 16; }
__host__ __device__ inline int fwd_window(int len2) { return kFwdOffsets + 2 + len2 + 16; }

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float lane_shl1(float v) {
  // lane l receives lane l + 1's value; lane 63 receives 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
typedef __attribute__((address_space(3))) const float lds_cfloat;

__global__ __launch_bounds__(256) void psa_fwd(ResolveParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  __builtin_amdgcn_s_setprio(3);  // short dependent chains: win VALU arbitration against the enumeration waves
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  PSA_TS_DECL()
  PSA_TS()
  FwdBlock fb;
  fb.seq = uni(P.fwd_blocks[blockIdx.x].seq);
  fb.s0 = uni(P.fwd_blocks[blockIdx.x].s0);
  const SeqDesc d = load_desc(P.desc + fb.seq);
  const int L2 = d.len2;
  int *rowb = (int *)(smem + kDevTable * 4);        // fwd_rows(L2) ints, 16-byte aligned
  int *aw = rowb + fwd_rows(L2);                    // fwd_window(L2) byte offsets
  uint32_t *red = (uint32_t *)(aw + fwd_window(L2));
  PSA_TS()
  for (int i = tid; i < kDevTable; i += 256) ((float *)smem)[i] = P.table_t[i];
  // The residues come from global memory eight per thread at a time, all eight requested before
  // the first is used: one memory round trip per 2048 positions instead of one per 256.
  constexpr int B = 8;
  {
    // a row entry is the LDS address of the table row of that Seq2 residue, so that a table
    // entry's address is one add: row + 4 * (Seq1 code)
    const int tbase = (int)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const unsigned char *src = P.seq2_pool + d.off2;
    const int n = fwd_rows(L2);
    for (int i0 = tid; i0 < n; i0 += 256 * B) {
      int code[B];
#pragma unroll
      for (int u = 0; u < B; ++u) code[u] = src[min(i0 + 256 * u, L2 - 1)];
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int i = i0 + 256 * u;
        if (i < n) rowb[i] = tbase + (i < L2 ? code[u] : kPadCode) * (kRowStride * 4);
      }
    }
  }
  {
    const int64_t win0 = (int64_t)fb.s0 - 2;  // Seq1 index of window entry 0
    const int n = fwd_window(L2);
    for (int i0 = tid; i0 < n; i0 += 256 * B) {
      int code[B];
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int64_t idx = win0 + i0 + 256 * u;
        code[u] = P.seq1[idx < 0 ? 0 : (idx < P.len1 ? idx : P.len1 - 1)];
      }
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int i = i0 + 256 * u;
        const int64_t idx = win0 + i;
        if (i < n) aw[i] = 4 * ((idx >= 0 && idx < P.len1) ? code[u] : kPadCode);
      }
    }
  }
  __syncthreads();
  PSA_TS()
  const int s = fb.s0 + 62 * w + lane;  // the offset whose c this lane holds (its b: s - 1, its a: s - 2)
  const int *ap = aw + 62 * w + lane;   // ap[j] = 4 * code of Seq1[s - 2 + j]
  const float NEG = -INFINITY;
  float a = 0.0f;
  f2 bc = {NEG, NEG};
#define PSA_T(addr) (*(lds_cfloat *)(uintptr_t)(unsigned)(addr))
#define PSA_STEP(x0)                                             \
  {                                                              \
    const f2 t_ = bc + (f2){(x0), (x0)};                         \
    const float nc_ = __builtin_fmaxf(lane_shl1(bc.x), t_.y);    \
    const float nb_ = __builtin_fmaxf(lane_shl1(a), t_.x);       \
    a = a + (x0);                                                \
    bc = (f2){nb_, nc_};                                         \
  }
  // groups of four steps in which all three states advance: j + 3 <= L2 - 3.  Two groups per
  // iteration, so the table entries alternate between two register sets without copies.
  const int ngroups = (L2 - 2) / 4;
  const int4 *rows4 = (const int4 *)rowb;
  float xa[4], xb[4];
  int4 o, r;
  {
    const int4 r0 = rows4[0];
    xa[0] = PSA_T(r0.x + ap[0]);
    xa[1] = PSA_T(r0.y + ap[1]);
    xa[2] = PSA_T(r0.z + ap[2]);
    xa[3] = PSA_T(r0.w + ap[3]);
  }
  o = make_int4(ap[4], ap[5], ap[6], ap[7]);
  r = rows4[1];
  int g = 0;
  for (; g + 2 <= ngroups; g += 2) {
    // table entries of group g+1 (offsets fetched one iteration ago), offsets of group g+2
    xb[0] = PSA_T(r.x + o.x);
    xb[1] = PSA_T(r.y + o.y);
    xb[2] = PSA_T(r.z + o.z);
    xb[3] = PSA_T(r.w + o.w);
    int jn = 4 * g + 8;
    o = make_int4(ap[jn], ap[jn + 1], ap[jn + 2], ap[jn + 3]);
    r = rows4[g + 2];
    PSA_STEP(xa[0]) PSA_STEP(xa[1]) PSA_STEP(xa[2]) PSA_STEP(xa[3])
    // group g+2 and g+3 likewise
    xa[0] = PSA_T(r.x + o.x);
    xa[1] = PSA_T(r.y + o.y);
    xa[2] = PSA_T(r.z + o.z);
    xa[3] = PSA_T(r.w + o.w);
    jn += 4;
    o = make_int4(ap[jn], ap[jn + 1], ap[jn + 2], ap[jn + 3]);
    r = rows4[g + 3];
    PSA_STEP(xb[0]) PSA_STEP(xb[1]) PSA_STEP(xb[2]) PSA_STEP(xb[3])
  }
  if (g < ngroups) {  // xa holds group g
    PSA_STEP(xa[0]) PSA_STEP(xa[1]) PSA_STEP(xa[2]) PSA_STEP(xa[3])
  }
  // remaining positions, one at a time
  for (int j = 4 * ngroups; j < L2; ++j) {
    const float x0 = PSA_T(rowb[j] + ap[j]);
    const f2 t = bc + (f2){x0, x0};
    const float nc = __builtin_fmaxf(lane_shl1(bc.x), t.y);
    const float nb = (j <= L2 - 2) ? __builtin_fmaxf(lane_shl1(a), t.x) : NEG;  // one deletion must still follow
    if (j <= L2 - 3) a = a + x0;                                               // two deletions must still follow
    bc = (f2){nb, nc};
  }
#undef PSA_STEP
#undef PSA_T
  PSA_TS()
  const float c = bc.y;
  uint32_t key = 0u;
  if (lane < 62 && s < d.s_end) {
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
#ifdef PSA_TRACE
  PSA_TS()
  if (tid == 0 && blockIdx.x < 6)
    printf("fwd blk %d L2 %d: desc %llu stage %llu chain %llu tail %llu\n", (int)blockIdx.x, L2, ts_[1] - ts_[0], ts_[2] - ts_[1],
           ts_[3] - ts_[2], ts_[4] - ts_[3]);
#endif
}

// pass 2: list the offsets whose maximum equals the Seq2's maximum
__global__ __launch_bounds__(256) void psa_collect(ResolveParams P) {
  const FwdBlock fb = P.fwd_blocks[blockIdx.x];
  const SeqDesc d = P.desc[fb.seq];
  const int s = fb.s0 + threadIdx.x;
  if (threadIdx.x < kFwdOffsets && s < d.s_end) {
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

// ---------------------------------------------------------------------------------
// passes 3 and 4 work on one offset that attains the maximum.  LDS holds the table
// entries Seq2 position j meets there after 0 deletions (d0[j]) and after 2 and 1
// deletions (d21[j] = {d2, d1}).  d21[L2-1].y is -inf: a chain with one deletion
// cannot consume the last position.  Both arrays are padded with zeros (d21 with
// 256 + kTiePad pairs) so that lanes and prefetches may run past the end: adding +0.0
// changes no value.
// ---------------------------------------------------------------------------------
constexpr int kTiePad = 64;
__host__ __device__ inline int tie_d0_len(int len2) { return ((len2 + kTiePad + 3) / 4) * 4; }
__host__ __device__ inline int tie_d21_len(int len2) { return len2 + 256 + kTiePad; }

// The residues come from global memory eight per thread at a time, all requested before the first
// is used (one memory round trip per 2048 positions), into two byte arrays in LDS behind d21: Seq2,
// and the window of Seq1 that starts two positions before the offset; the table entries are then
// looked up from LDS alone.  Ends with a barrier.
__host__ __device__ inline int tie_codes_len(int len2) { return ((len2 + 2 + 15) / 16) * 16; }

__device__ __forceinline__ void fill_diagonals(float *d0, f2 *d21, const float *T, const ResolveParams &P,
                                               const SeqDesc &d, int s) {
  const int L2 = d.len2, tid = threadIdx.x;
  unsigned char *c2 = (unsigned char *)(d21 + tie_d21_len(L2));  // Seq2[j]
  unsigned char *c1 = c2 + tie_codes_len(L2);                    // Seq1[s - 2 + i], i < L2 + 2
  constexpr int B = 8;
  const unsigned char *src2 = P.seq2_pool + d.off2;
  for (int i0 = tid; i0 < L2 + 2; i0 += 256 * B) {
    int a[B], b[B];
#pragma unroll
    for (int u = 0; u < B; ++u) {
      const int i = i0 + 256 * u;
      const int64_t idx = (int64_t)s - 2 + i;
      b[u] = src2[min(i, L2 - 1)];
      a[u] = P.seq1[idx < 0 ? 0 : (idx < P.len1 ? idx : P.len1 - 1)];
    }
#pragma unroll
    for (int u = 0; u < B; ++u) {
      const int i = i0 + 256 * u;
      const int64_t idx = (int64_t)s - 2 + i;
      if (i < L2) c2[i] = (unsigned char)b[u];
      if (i < L2 + 2) c1[i] = (unsigned char)((idx >= 0 && idx < P.len1) ? a[u] : kPadCode);
    }
  }
  __syncthreads();
  // (four positions per thread in flight: the three dependent LDS reads of a position are latency)
  constexpr int U = 4;
  for (int j0 = tid; j0 < tie_d21_len(L2); j0 += 256 * U) {
    int cb[U], ca[U][3];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = min(j0 + 256 * u, L2 - 1);
      cb[u] = c2[j];
      ca[u][0] = c1[j + 2];
      ca[u][1] = c1[j + 1];
      ca[u][2] = c1[j];
    }
    float t[U][3];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float *row = T + cb[u] * kRowStride;
      t[u][0] = row[ca[u][0]];
      t[u][1] = row[ca[u][1]];
      t[u][2] = row[ca[u][2]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + 256 * u;
      if (j < tie_d21_len(L2)) {
        const bool in = j < L2;
        if (j < tie_d0_len(L2)) d0[j] = in ? t[u][0] : 0.0f;
        d21[j] = (f2){in ? t[u][2] : 0.0f, in ? (j == L2 - 1 ? -INFINITY : t[u][1]) : 0.0f};
      }
    }
  }
  __syncthreads();
}

// Values requested per group in the chains below (two groups of registers take turns).
#ifndef PSA_TIE_GROUP
#define PSA_TIE_GROUP 16
#endif
constexpr int kG = PSA_TIE_GROUP;
static_assert(kG == 8 || kG == 16, "group of 8 or 16");

// Serial float32 chains over LDS arrays.  The adds are dependent by nature (the reference's
// rounding order) and a lone wave issues a dependent add every 10.5 cycles; what can be hidden
// is the LDS latency: the next kG values are requested before the current kG are added (the
// scheduling barriers keep the compiler from sinking the requests below the adds, or from
// hoisting them into the adds, which costs a second set of registers and copies).
//
// v += src[STRIDE * j] for j in [from, to), left to right, every lane the same value.  With HAND
// lane `my` stops before position `my`: it adds +0.0 from there on (which changes no value), so
// that it ends with the value of the chain before position `my` -- the compare and the select
// that zero an element do not depend on the chain, only the add does.
// Reads up to 2 kG elements past `to` (the arrays are padded).
template <int STRIDE, bool HAND>
__device__ __forceinline__ float add_run(const float *src, int from, int to, float v, int my) {
  int n = to - from;
  if (n <= 0) return v;
  int j = from;
  int rel = my - from;  // positions this lane still adds
  float y[kG], z[kG];
#define PSA_LOAD(r, j0)                                                  \
  _Pragma("unroll") for (int k = 0; k < kG; ++k) r[k] = src[STRIDE * ((j0) + k)];
#define PSA_ADDS(r)                                                       \
  _Pragma("unroll") for (int k = 0; k < kG; ++k) v += HAND ? (k < rel ? r[k] : 0.0f) : r[k]; \
  rel -= kG;
#define PSA_ADDS_PART(r, j0)                                              \
  _Pragma("unroll") for (int k = 0; k < kG; ++k)                          \
    if ((j0) + k < to) v += HAND ? (k < rel ? r[k] : 0.0f) : r[k];       \
  rel -= kG;
  PSA_LOAD(y, j)
  while (n >= 2 * kG) {
    PSA_LOAD(z, j + kG)
    __builtin_amdgcn_sched_barrier(0);
    PSA_ADDS(y)
    __builtin_amdgcn_sched_barrier(0);
    PSA_LOAD(y, j + 2 * kG)
    __builtin_amdgcn_sched_barrier(0);
    PSA_ADDS(z)
    j += 2 * kG;
    n -= 2 * kG;
  }
  if (n >= kG) {
    PSA_LOAD(z, j + kG)
    __builtin_amdgcn_sched_barrier(0);
    PSA_ADDS(y)
    PSA_ADDS_PART(z, j + kG)
  } else {
    PSA_ADDS_PART(y, j)
  }
#undef PSA_LOAD
#undef PSA_ADDS
#undef PSA_ADDS_PART
  return v;
}

// pass 3: lane p computes max_q score(s,p,q) exactly: the chain that deleted p starts
// from a_p (the undeleted prefix sum) and then runs the same monotone recurrence,
//   c' = max(v, c + d2[j]),  v' = v + d1[j]   for j = p+1 .. L2-1.
// Lanes are skewed in time (lane p reads position p + t at step t), so every lane is in
// the same phase and no per-lane selects are needed.  The smallest p that attains the
// maximum wins.
// Two positions are taken per iteration: with (c, v) in one register pair,
//   {t0, v1} = {c, v} + {d2[j], d1[j]},   {t1, v2} = {t0, v1} + {d2[j+1], d1[j+1]}
// are two packed adds, and  c'' = max(v1, c' + d2[j+1]) = max(v1, fl(v + d2[j+1]), t1)  because
// fl(max(x, y) + d) = max(fl(x + d), fl(y + d)): the dependent chain through c is add, add, max3
// per two positions instead of add, max per position, and the values are the same bit for bit.
__global__ __launch_bounds__(256) void psa_first_p(ResolveParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  __builtin_amdgcn_s_setprio(3);  // short dependent chains: win VALU arbitration against the enumeration waves
  PSA_TS_DECL()
  PSA_TS()
  __shared__ int skip_flag[1];
  float *T = (float *)smem;
  const int tid = threadIdx.x;
  for (int i = tid; i < kDevTable; i += 256) T[i] = P.table_t[i];
  const unsigned nact = *P.nactive;
  const long long nitems = (long long)nact * P.pchunks;
  PSA_TS()
  // Items in slice-major order: every active offset's first 256 positions, then every offset's second 256,
  // and so on, so that a block's items never go back to a lower slice.  With thousands of offsets tied at the
  // maximum (low-entropy inputs: every offset of an all-same pair) the first items to finish find a small p,
  // nearly every other item is skipped, and in a one-Seq2 batch a block that meets a slice at or above the
  // known p is done with all its items.
  for (long long item = blockIdx.x; item < nitems; item += gridDim.x) {
    ActiveOffset ao;
    const long long ai = item % nact;
    ao.seq = uni(P.active[ai].seq);
    ao.s = uni(P.active[ai].s);
    const int pc = (int)(item / nact);
    // (requested together with the descriptor: one memory round trip, not three)
    // (an agent-scope load: a plain one can be served from this CU's vector cache, which other CUs' atomics
    // never refresh -- a block whose own items find nothing would go on seeing the value it read first)
    const int known_p = tid == 0 ? __hip_atomic_load(&P.acc[ao.seq].res_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const float fstar = key2f(P.acc[ao.seq].res_key);
    const SeqDesc d = load_desc(P.desc + ao.seq);
    const int L2 = d.len2;
    const int p0 = pc * 256;
    if (p0 > L2 - 2) continue;
    __syncthreads();
    if (tid == 0) skip_flag[0] = p0 >= known_p;  // a p at or below this slice's first is already known
    __syncthreads();
    if (uni(skip_flag[0])) {  // one decision for the whole block
      if (P.one_seq) break;   // later items of this block sit in the same or a higher slice of the same Seq2
      continue;
    }
    float *d0 = T + kDevTable;
    f2 *d21 = (f2 *)(d0 + tie_d0_len(L2));
    PSA_TS()
    fill_diagonals(d0, d21, T, P, d, ao.s);
    PSA_TS()
    const int p = p0 + tid;
    // a_p: prefix of the undeleted chain; lane p keeps the value before position p
    float v = add_run<1, false>(d0, 0, p0, 0.0f, 0);
    PSA_TS()
    const int pend = min(p0 + 256, L2 - 1);  // p <= L2 - 2
    const float mine = add_run<1, true>(d0, p0, pend, v, p);
    PSA_TS()
    // skewed chain: step t handles position p + 1 + t; whole groups of kG (the padding is neutral:
    // v is -inf from position L2 - 1 on, and c + 0 = c)
    const f2 *dp = d21 + p + 1;
    const int nsteps = (L2 - 1 - p0 + kG - 1) & ~(kG - 1);
    f2 cv = {-INFINITY, mine};
    f2 y[kG], z[kG];
#define PSA_LOAD(r, t0) _Pragma("unroll") for (int k = 0; k < kG; ++k) r[k] = dp[(t0) + k];
#define PSA_STEPS(r)                                                              \
  _Pragma("unroll") for (int k = 0; k < kG; k += 2) {                             \
    const f2 a0 = cv + r[k];                                                      \
    f2 a1 = a0 + r[k + 1];                                                        \
    const float u = cv.y + r[k + 1].x;                                            \
    a1.x = __builtin_fmaxf(__builtin_fmaxf(a0.y, u), a1.x);                       \
    cv = a1;                                                                      \
  }
    PSA_LOAD(y, 0)
    int t = 0;
    for (; t + 2 * kG <= nsteps; t += 2 * kG) {
      PSA_LOAD(z, t + kG)
      __builtin_amdgcn_sched_barrier(0);
      PSA_STEPS(y)
      __builtin_amdgcn_sched_barrier(0);
      PSA_LOAD(y, t + 2 * kG)
      __builtin_amdgcn_sched_barrier(0);
      PSA_STEPS(z)
    }
    if (t < nsteps) {
      PSA_STEPS(y)
    }
#undef PSA_LOAD
#undef PSA_STEPS
    PSA_TS()
    const float c = cv.x;
    int cand = (p <= L2 - 2 && c == fstar) ? p : INT_MAX;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) cand = min(cand, __shfl_xor(cand, m));
    if ((tid & 63) == 0 && cand != INT_MAX) atomicMin(&P.acc[ao.seq].res_p, cand);
#ifdef PSA_TRACE
    PSA_TS()
    if (tid == 0 && blockIdx.x < 8)
      printf("first_p blk %d p0 %d L2 %d: nact %llu desc %llu fill %llu prefix %llu hand %llu main %llu tail %llu\n", (int)blockIdx.x, p0,
             L2, ts_[1] - ts_[0], ts_[2] - ts_[1], ts_[3] - ts_[2], ts_[4] - ts_[3], ts_[5] - ts_[4], ts_[6] - ts_[5], ts_[7] - ts_[6]);
    nts_ = 2;
#endif
  }
}

// pass 4: with p* fixed, lane q evaluates score(s,p*,q) exactly as the reference adds
// it: the one-deletion chain value before position q, then d2[j] for j > q (skewed in
// time like pass 3).  The smallest q, then the smallest offset, wins.
//
// ROWS: the same evaluation for the rows of the enumeration stage's shortlist instead of the
// offsets that attain the float32 maximum -- p comes from the row, only the blocks of q the row
// names are visited, and the row keeps max_q of (score image, ~q): the highest score of its
// candidates and the smallest q that attains it; the Seq2 keeps the max of (score image, ~p, ~q).
template <bool ROWS>
__device__ __forceinline__ void first_q_body(const ResolveParams &P) {
  extern __shared__ __align__(16) unsigned char smem[];
  __builtin_amdgcn_s_setprio(3);  // short dependent chains: win VALU arbitration against the enumeration waves
  PSA_TS_DECL()
  PSA_TS()
  __shared__ int skip_flag[1];
  float *T = (float *)smem;
  const int tid = threadIdx.x;
  for (int i = tid; i < kDevTable; i += 256) T[i] = P.table_t[i];
  const unsigned nact = ROWS ? min(*P.short_total, (unsigned)P.short_rows_cap) : *P.nactive;
  const long long nitems = (long long)nact * P.pchunks;
  PSA_TS()
  for (long long item = blockIdx.x; item < nitems; item += gridDim.x) {
    ActiveOffset ao;
    // (the tie offsets in slice-major order, like psa_first_p; the shortlist's rows row by row)
    const int qc = ROWS ? (int)(item % P.pchunks) : (int)(item / nact);
    unsigned long long known = ~0ull;
    float fstar = 0.0f;
    int pstar;
    ShortRow *row = nullptr;
    if (ROWS) {
      // (everything read here is final: the kernel that lists the rows has finished)
      row = P.short_rows + item / P.pchunks;
      if (!uni((int)((row->qmask >> qc) & 1u))) continue;
      ao.seq = uni(row->seq);
      ao.s = uni(row->s);
      pstar = uni((int)row->p);
      if (uni((int)(P.eacc[ao.seq].short_count > (unsigned)P.short_quota))) continue;  // given up
    } else {
      ao.seq = uni(P.active[item % nact].seq);
      ao.s = uni(P.active[item % nact].s);
      // (requested together with the descriptor: one memory round trip, not three)
      // (an agent-scope load, as in psa_first_p: the pruning below lives on other blocks' results)
      known = tid == 0 ? __hip_atomic_load(&P.acc[ao.seq].res_loc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
      fstar = key2f(P.acc[ao.seq].res_key);
      pstar = uni(P.acc[ao.seq].res_p);
    }
    const SeqDesc d = load_desc(P.desc + ao.seq);
    const int L2 = d.len2;
    const int q0 = qc * 256;
    if (pstar == INT_MAX || q0 > L2 - 1 || q0 + 255 <= pstar) continue;
    __syncthreads();
    if (!ROWS) {
      // 2: a q below this slice is already known -- this item and, in a one-Seq2 batch, every later item of the
      // block is done; 1: the smallest q there is, p* + 1, is known at a smaller offset -- this offset cannot win
      if (tid == 0)
        skip_flag[0] = known == ~0ull ? 0
                       : (int)(known >> 32) < q0 ? 2
                       : ((int)(known >> 32) == pstar + 1 && (int)(uint32_t)known < ao.s) ? 1 : 0;
      __syncthreads();
      const int skip = uni(skip_flag[0]);  // one decision for the whole block
      if (skip == 2 && P.one_seq) break;
      if (skip) continue;
    }
    float *d0 = T + kDevTable;
    f2 *d21 = (f2 *)(d0 + tie_d0_len(L2));
    PSA_TS()
    fill_diagonals(d0, d21, T, P, d, ao.s);
    PSA_TS()
    const int q = q0 + tid;
    const float *d1 = (const float *)d21 + 1;  // stride 2
    float v = add_run<1, false>(d0, 0, pstar, 0.0f, 0);  // a_{p*}
    int j = pstar + 1;
    if (j < q0) {  // positions every lane of the block consumes
      v = add_run<2, false>(d1, j, q0, v, 0);
      j = q0;
    }
    PSA_TS()
    // the value handed over when q is deleted: the chain before position q
    const float mine = add_run<2, true>(d1, j, min(q0 + 256, L2), v, q);
    PSA_TS()
    // skewed chain over d2, whole groups of kG (the padding adds +0.0)
    const float *dp = (const float *)(d21 + q + 1);
    const int nsteps = (L2 - 1 - q0 + kG - 1) & ~(kG - 1);
    float c = mine;
    float y[kG], z[kG];
#define PSA_LOAD(r, t0) _Pragma("unroll") for (int k = 0; k < kG; ++k) r[k] = dp[2 * ((t0) + k)];
#define PSA_ADDS(r) _Pragma("unroll") for (int k = 0; k < kG; ++k) c += r[k];
    PSA_LOAD(y, 0)
    int t = 0;
#define PSA_TWO_GROUPS()                   \
  PSA_LOAD(z, t + kG)                      \
  __builtin_amdgcn_sched_barrier(0);       \
  PSA_ADDS(y)                              \
  __builtin_amdgcn_sched_barrier(0);       \
  PSA_LOAD(y, t + 2 * kG)                      \
  __builtin_amdgcn_sched_barrier(0);       \
  PSA_ADDS(z)                              \
  __builtin_amdgcn_sched_barrier(0);       \
  t += 2 * kG;
    // (4 kG positions per iteration: the kG values carried into the next iteration are copied once)
    while (t + 4 * kG <= nsteps) {
      PSA_TWO_GROUPS()
      PSA_TWO_GROUPS()
    }
    if (t + 2 * kG <= nsteps) {
      PSA_TWO_GROUPS()
    }
#undef PSA_TWO_GROUPS
    if (t < nsteps) {
      PSA_ADDS(y)
    }
#undef PSA_LOAD
#undef PSA_ADDS
    PSA_TS()
    if (ROWS) {
      unsigned long long k = 0ull;
      if (q > pstar && q <= L2 - 1) k = ((unsigned long long)f2key(c) << 32) | (0xffffffffu - (uint32_t)q);
      k = wave_max_u64(k);
      if ((tid & 63) == 0 && k != 0ull) {
        atomicMax(&row->best, k);
        const unsigned long long qmin = 0xffffffffu - (uint32_t)k;
        atomicMax(&P.eacc[ao.seq].short_best, (k & 0xffffffff00000000ull) | ((unsigned long long)(0xffffu - (uint32_t)pstar) << 16) |
                                                  (0xffffull - qmin));
      }
    } else {
      unsigned long long cand = ~0ull;
      if (q > pstar && q <= L2 - 1 && c == fstar) cand = ((unsigned long long)(uint32_t)q << 32) | (uint32_t)ao.s;
      cand = wave_min_u64(cand);
      if ((tid & 63) == 0 && cand != ~0ull) atomicMin(&P.acc[ao.seq].res_loc, cand);
    }
#ifdef PSA_TRACE
    PSA_TS()
    if (tid == 0 && blockIdx.x < 8)
      printf("first_q blk %d q0 %d pstar %d L2 %d: nact %llu desc %llu fill %llu prefix %llu hand %llu main %llu tail %llu\n", (int)blockIdx.x, q0,
             pstar, L2, ts_[1] - ts_[0], ts_[2] - ts_[1], ts_[3] - ts_[2], ts_[4] - ts_[3], ts_[5] - ts_[4], ts_[6] - ts_[5], ts_[7] - ts_[6]);
    nts_ = 2;
#endif
  }
}

__global__ __launch_bounds__(256) void psa_first_q(ResolveParams P) { first_q_body<false>(P); }
__global__ __launch_bounds__(256) void psa_short_rows(ResolveParams P) { first_q_body<true>(P); }

// a row's best candidate as the key its Seq2 keeps the maximum of: (score, smallest p, smallest q)
__device__ __forceinline__ unsigned long long short_row_key(const ShortRow &r) {
  const unsigned long long qmin = 0xffffffffu - (uint32_t)r.best;
  return (r.best & 0xffffffff00000000ull) | ((unsigned long long)(0xffffu - r.p) << 16) | (0xffffull - qmin);
}

// The smallest offset among the rows whose best candidate is their Seq2's best (score, p, q).
__global__ __launch_bounds__(256) void psa_short_pick(ResolveParams P) {
  const unsigned n = min(*P.short_total, (unsigned)P.short_rows_cap);
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const ShortRow r = P.short_rows[i];
    if (r.best == 0ull) continue;
    SeqAcc *a = P.eacc + r.seq;
    if (a->short_count <= (unsigned)P.short_quota && short_row_key(r) == a->short_best) atomicMin(&a->short_s, (uint32_t)r.s);
  }
}

// ---------------------------------------------------------------------------------
// result records: decode the accumulators and count residue classes of the winners
// ---------------------------------------------------------------------------------
// (the class table sits in LDS; a thread fetches the residues of up to 8 positions before it
// looks any of them up, so the global-memory latency is paid once, not once per position)
__device__ void count_classes(const FinalParams &P, const SeqDesc &d, int s, int p, int q,
                              const unsigned char *lds_class, int *lds_counts, int32_t *out) {
  const int tid = threadIdx.x;
  if (tid < 4) lds_counts[tid] = 0;
  __syncthreads();
  int local[4] = {0, 0, 0, 0};
  const int m = d.len2 - 2;
  constexpr int U = 8;
  const unsigned char *s1 = P.seq1 + (int64_t)s, *s2 = P.seq2_pool + d.off2;
  for (int i0 = tid; i0 < m; i0 += U * 256) {
    int a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = min(i0 + u * 256, m - 1);
      const int j = i < p ? i : (i < q - 1 ? i + 1 : i + 2);
      a[u] = s1[i];
      b[u] = s2[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int cls = i0 + u * 256 < m ? (int)lds_class[b[u] * kRowStride + a[u]] : -1;
      local[0] += cls == 0;
      local[1] += cls == 1;
      local[2] += cls == 2;
      local[3] += cls == 3;
    }
  }
  for (int k = 0; k < 4; ++k)
    if (local[k]) atomicAdd(&lds_counts[k], local[k]);
  __syncthreads();
  if (tid < 4) out[tid] = lds_counts[tid];
  __syncthreads();
}

// P.which says which half of the records this launch fills: the float32 (resolve) fields,
// the fixed-point (enumeration) fields, or both.  The halves are launched on the streams
// that produce their inputs, so the resolve half is done long before the enumeration ends.
__global__ __launch_bounds__(256) void psa_finalize(FinalParams P) {
  __shared__ int lds_counts[4];
  __shared__ unsigned lds_pick;
  __shared__ __align__(4) unsigned char lds_class[kDevTable];
  const int i = blockIdx.x;
  static_assert(kDevTable % 4 == 0 && kDevTable / 4 <= 256, "one 4-byte load per thread");
  if (threadIdx.x < kDevTable / 4) ((uint32_t *)lds_class)[threadIdx.x] = ((const uint32_t *)P.class_t)[threadIdx.x];
  const SeqDesc d = P.desc[i];
  const SeqAcc a = P.acc[i];
  SeqAcc ea = P.eacc[i];
  psa_result *r = P.out + i;
  const int tid = threadIdx.x;
  if (uni((int)((P.which & PSA_STAGE_ENUMERATE) && P.pick_rows != nullptr))) {
    // this Seq2's smallest offset among the rows whose best candidate is its best (every block reads the whole
    // list: a small batch's)
    if (tid == 0) lds_pick = ~0u;
    __syncthreads();
    if (ea.short_count <= (unsigned)P.short_quota && ea.short_best != 0ull) {
      const unsigned n = min(*P.short_total, (unsigned)P.pick_rows_cap);
      unsigned best_s = ~0u;
      for (unsigned k = tid; k < n; k += 256u) {
        const ShortRow row = P.pick_rows[k];
        if (row.seq == i && row.best != 0ull && short_row_key(row) == ea.short_best) best_s = min(best_s, (uint32_t)row.s);
      }
      if (best_s != ~0u) atomicMin(&lds_pick, best_s);
    }
    __syncthreads();
    ea.short_s = lds_pick;
    if (tid == 0) {
      // every block reads the row counter: the last one to have done so resets it for the next run
      __threadfence();
      if (atomicAdd(P.pick_done, 1u) == gridDim.x - 1u) {
        *P.short_total = 0u;
        *P.pick_done = 0u;
      }
    }
  }
  // Each half is the last reader of its accumulators in a run and leaves them reset for the
  // next one (what psa_init_acc writes), so a resident batch needs no set-up kernel per run.
  __syncthreads();
  if (tid == 0) {
    SeqAcc *w = P.acc + i;
    // (only what this run's stages used: a run without a stage clears that half of the records
    // but must not touch accumulators that a run enqueued behind it may already be filling)
    if (P.which & P.stages & PSA_STAGE_RESOLVE) {
      w->res_loc = ~0ull;
      w->res_key = 0u;
      w->res_p = INT_MAX;
      if (i == 0) *P.nactive = 0u;
    }
    if (P.which & P.stages & PSA_STAGE_ENUMERATE) {
      P.eacc[i].enum_key = 0ull;
      P.eacc[i].enum_loc = ~0ull;
      P.eacc[i].short_best = 0ull;
      P.eacc[i].short_s = ~0u;
      P.eacc[i].short_count = 0u;
      // (its readers ran before this kernel on the same stream -- unless the blocks of this kernel read it)
      if (i == 0 && P.short_total && !P.pick_rows) *P.short_total = 0u;
    }
  }
  if (uni((int)(P.which & PSA_STAGE_RESOLVE))) {
    const bool res_ok = uni((P.stages & PSA_STAGE_RESOLVE) && a.res_loc != ~0ull && a.res_p != INT_MAX);
    if (res_ok) {
      const int p = a.res_p, q = (int)(a.res_loc >> 32), s = (int)(uint32_t)a.res_loc;
      count_classes(P, d, s, p, q, lds_class, lds_counts, r->counts);
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
  }
  if (uni((int)(P.which & PSA_STAGE_ENUMERATE))) {
    const bool enum_ok = uni((P.stages & PSA_STAGE_ENUMERATE) && d.frac_bits >= 0 && ea.enum_loc != ~0ull);
    if (enum_ok) {
      const int p = (int)(0xffffffffu - (uint32_t)ea.enum_key);
      const int q = (int)(ea.enum_loc >> 32), s = (int)(uint32_t)ea.enum_loc;
      count_classes(P, d, s, p, q, lds_class, lds_counts, r->enum_counts);
      if (tid == 0) {
        r->enum_n = p + 1;
        r->enum_k = q + 1;
        r->enum_offset = s;
        r->enum_score = (int)((uint32_t)(ea.enum_key >> 32) ^ 0x80000000u);
      }
    } else if (tid == 0) {
      r->enum_n = r->enum_k = r->enum_offset = r->enum_score = 0;
      r->enum_counts[0] = r->enum_counts[1] = r->enum_counts[2] = r->enum_counts[3] = 0;
    }
    // The replayed shortlist of this Seq2: its winner is the highest float32 score, ties to the
    // smallest (n, k, offset) -- the order in which the reference keeps the first maximum
    // (S/main.c:44-83: mutants outer, offsets inner, strict >).
    if (P.short_out) {
      ShortOut *so = P.short_out + i;
      const bool listed = uni(enum_ok && P.short_total && d.enum_slack >= 0);
      const bool within = uni(listed && ea.short_count <= (unsigned)P.short_quota);
      const bool complete = uni(within && ea.short_best != 0ull && ea.short_s != ~0u);
      if (within && !complete && P.short_empty_ok) {
        // one shard of a long pair, window hung below the maximum over all shards: this shard holds no
        // candidate inside it -- a complete list without a winner
        if (tid == 0) {
          so->score_bits = 0u;
          so->n = so->k = so->offset = 0;
          so->counts[0] = so->counts[1] = so->counts[2] = so->counts[3] = 0;
          so->count = 0;
          so->state = 1;
        }
      } else if (complete) {
        const int p = (int)(0xffffu - (uint32_t)((ea.short_best >> 16) & 0xffffu));
        const int q = (int)(0xffffu - (uint32_t)(ea.short_best & 0xffffu));
        const int s = (int)ea.short_s;
        // (the replayed winner is nearly always the enumerated one, whose counts are still in LDS)
        const bool same = uni((int)(p == (int)(0xffffffffu - (uint32_t)ea.enum_key) && q == (int)(ea.enum_loc >> 32) &&
                                    s == (int)(uint32_t)ea.enum_loc));
        if (same) {
          if (tid < 4) so->counts[tid] = lds_counts[tid];
        } else {
          count_classes(P, d, s, p, q, lds_class, lds_counts, so->counts);
        }
        if (tid == 0) {
          so->score_bits = __float_as_uint(key2f((uint32_t)(ea.short_best >> 32)));
          so->n = p + 1;
          so->k = q + 1;
          so->offset = s;
          so->count = (int)ea.short_count;
          so->state = 1;
        }
      } else if (tid == 0) {
        so->score_bits = 0u;
        so->n = so->k = so->offset = 0;
        so->counts[0] = so->counts[1] = so->counts[2] = so->counts[3] = 0;
        so->count = listed ? (int)min(ea.short_count, (unsigned)P.short_quota) : 0;
        so->state = listed ? 2 : 0;
      }
    }
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

// Set-up of a small batch in one launch: the work lists staged in pinned host memory (read
// through its device mapping, 16 bytes per thread) go to their device block, the accumulators
// and counters get their initial values and the per-chunk keys are cleared -- instead of an
// asynchronous copy, a memset and psa_init_acc on two streams with an event between them.
__global__ __launch_bounds__(256) void psa_setup_small(const uint4 *src, uint4 *dst, int n16, SeqAcc *acc, int nacc,
                                                        unsigned *counters, int ncounters, unsigned long long *best,
                                                        int nbest) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16) dst[i] = src[i];
  if (i < nacc) {
    SeqAcc a;
    a.enum_key = 0ull;
    a.enum_loc = ~0ull;
    a.res_loc = ~0ull;
    a.res_key = 0u;
    a.res_p = INT_MAX;
    a.short_best = 0ull;
    a.short_s = ~0u;
    a.short_count = 0u;
    acc[i] = a;
  }
  if (i < ncounters) counters[i] = 0u;
  for (int k = i; k < nbest; k += gridDim.x * blockDim.x) best[k] = 0ull;
}

hipError_t launch_setup_small(const void *src, void *dst, size_t bytes, SeqAcc *acc, int nacc, unsigned *counters,
                              int ncounters, unsigned long long *best, size_t nbest, hipStream_t st) {
  const int n16 = (int)((bytes + 15) / 16);
  const int m = std::max(std::max(n16, nacc), std::max(ncounters, (int)std::min<size_t>(nbest, 1 << 16)));
  hipLaunchKernelGGL(psa_setup_small, dim3((m + 255) / 256), dim3(256), 0, st, (const uint4 *)src, (uint4 *)dst, n16, acc,
                     nacc, counters, ncounters, best, (int)nbest);
  return hipGetLastError();
}

hipError_t launch_init(SeqAcc *acc, int n, unsigned *counters, int ncounters, hipStream_t st) {
  int m = n > ncounters ? n : ncounters;
  hipLaunchKernelGGL(psa_init_acc, dim3((m + 255) / 256), dim3(256), 0, st, acc, n, counters, ncounters);
  return hipGetLastError();
}

static size_t fwd_lds(int max_len2) {
  return (size_t)kDevTable * 4 + 4 * (size_t)fwd_rows(max_len2) + 4 * (size_t)fwd_window(max_len2) + 64;
}
static size_t tie_lds(int max_len2) {
  return (size_t)kDevTable * 4 + 4 * (size_t)tie_d0_len(max_len2) + 8 * (size_t)tie_d21_len(max_len2) +
         2 * (size_t)tie_codes_len(max_len2);
}

// The dynamic-LDS limits of the resolve kernels for the device that is current: once per context
// (psa_init), sized for the longest Seq2 the library takes.  The attribute belongs to the
// (kernel, device) pair, so nothing about it is remembered process-wide.
hipError_t set_resolve_kernel_attributes() {
  hipError_t e;
  const int tb = (int)tie_lds(PSA_MAX_SEQ2);
  e = hipFuncSetAttribute((const void *)psa_first_p, hipFuncAttributeMaxDynamicSharedMemorySize, tb);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)psa_first_q, hipFuncAttributeMaxDynamicSharedMemorySize, tb);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)psa_short_rows, hipFuncAttributeMaxDynamicSharedMemorySize, tb);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void *)psa_fwd, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)fwd_lds(PSA_MAX_SEQ2));
}

hipError_t launch_resolve(const ResolveParams &P, int nfwd_blocks, int max_len2, int tie_grid, hipStream_t st,
                          hipEvent_t ev_start) {
  hipError_t e;
  const size_t tb = tie_lds(max_len2);
  hipExtLaunchKernelGGL(psa_fwd, dim3(nfwd_blocks), dim3(256), fwd_lds(max_len2), st, ev_start, nullptr, 0, P);
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

hipError_t launch_short_replay(const ResolveParams &P, int max_len2, int grid, bool pick, hipStream_t st) {
  if (!P.short_rows) return hipSuccess;
  hipLaunchKernelGGL(psa_short_rows, dim3(grid), dim3(256), tie_lds(max_len2), st, P);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if ((e = debug_sync("psa_short_rows")) != hipSuccess) return e;
  if (!pick) return hipSuccess;  // (psa_finalize picks: FinalParams::pick_rows)
  hipLaunchKernelGGL(psa_short_pick, dim3(std::max(1, std::min(1024, P.short_rows_cap / 256))), dim3(256), 0, st, P);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return debug_sync("psa_short_pick");
}

hipError_t launch_finalize(const FinalParams &P, int n, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
  hipExtLaunchKernelGGL(psa_finalize, dim3(n), dim3(256), 0, st, ev_start, ev_stop, 0, P);
  return hipGetLastError();
}


}  // namespace psa
