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
#include <stdio.h>
#include <stdlib.h>

#include <hip/hip_ext.h>

#include "psa_device.h"

namespace psa {

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
// float32 resolve, pass 1: per-offset maximum over all (n,k); lanes = offsets.
// State before Seq2 position j: a (no deletion yet), b (one), c (two).
//   a' = a + x0,  b' = max(a, b + x1),  c' = max(b, c + x2),   x_d = T[A[s+j-d]][B[j]]
// (delete j, or keep it after d deletions).  Rounding is monotone, so taking the max
// before continuing the chain is exact.
//
// Consecutive lanes hold consecutive offsets, so x1 of lane l is x0 of lane l - 1 and x2
// is x0 of lane l - 2 at the same step: one LDS table read per lane and step, the other
// two entries arrive through wave-wide lane shifts (DPP wave_shr:1).  Lanes 0 and 1 of a
// wave have no left neighbours; they run the two offsets below the wave's first and are
// not used, so a wave yields 62 offsets and a block kFwdOffsets = 248.
// The kernel runs one wave per SIMD and is bound by how fast that wave can issue, i.e.
// by instructions per step: the Seq1 codes and the Seq2 row offsets of a group of four
// steps are fetched two groups ahead, the table entries one group ahead, so no LDS
// latency sits on the dependent chain.
//
// LDS: table | row byte offsets of Seq2 (128 * code per position) | Seq1 window as byte
// offsets (4 * code)
// ---------------------------------------------------------------------------------
__host__ __device__ inline int fwd_rows(int len2) { return ((len2 + 3) / 4) * 4 + 16; }
__host__ __device__ inline int fwd_window(int len2) { return kFwdOffsets + 2 + len2 + 16; }

__device__ __forceinline__ float lane_shr1(float v) {
  // lane l receives lane l - 1's value; lane 0 receives 0
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xf, 0xf, true));
}
// max without the canonicalising self-max that fmaxf costs per operand (no NaN can occur here:
// finite table entries, and -inf only ever meets finite values)
__device__ __forceinline__ float max_f32(float x, float y) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
typedef __attribute__((address_space(3))) const float lds_cfloat;

__global__ __launch_bounds__(256) void psa_fwd(ResolveParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  __builtin_amdgcn_s_setprio(3);  // short dependent chains: win VALU arbitration against the enumeration waves
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  FwdBlock fb;
  fb.seq = uni(P.fwd_blocks[blockIdx.x].seq);
  fb.s0 = uni(P.fwd_blocks[blockIdx.x].s0);
  const SeqDesc d = load_desc(P.desc + fb.seq);
  const int L2 = d.len2;
  int *rowb = (int *)(smem + kDevTable * 4);        // fwd_rows(L2) ints, 16-byte aligned
  int *aw = rowb + fwd_rows(L2);                    // fwd_window(L2) byte offsets
  uint32_t *red = (uint32_t *)(aw + fwd_window(L2));
  for (int i = tid; i < kDevTable; i += 256) ((float *)smem)[i] = P.table_t[i];
  {
    // a row entry is the LDS address of the table row of that Seq2 residue, so that a table
    // entry's address is one add: row + 4 * (Seq1 code)
    const int tbase = (int)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const unsigned char *src = P.seq2_pool + d.off2;
    for (int i = tid; i < fwd_rows(L2); i += 256)
      rowb[i] = tbase + (i < L2 ? (int)src[i] : kPadCode) * (kRowStride * 4);
  }
  const int win0 = fb.s0 - 2;  // Seq1 index of window entry 0
  for (int i = tid; i < fwd_window(L2); i += 256) {
    int64_t idx = (int64_t)win0 + i;
    aw[i] = 4 * ((idx >= 0 && idx < P.len1) ? (int)P.seq1[idx] : kPadCode);
  }
  __syncthreads();
  const int s = fb.s0 + 62 * w + lane - 2;
  const int *ap = aw + 62 * w + lane;  // ap[j] = 4 * code of Seq1[s + j]
  const float NEG = -INFINITY;
  float a = 0.0f, b = NEG, c = NEG;
#define PSA_T(addr) (*(lds_cfloat *)(uintptr_t)(unsigned)(addr))
#define PSA_STEP(x0)                        \
  {                                         \
    const float x1_ = lane_shr1(x0);        \
    const float x2_ = lane_shr1(x1_);       \
    const float nc_ = max_f32(b, c + x2_);  \
    const float nb_ = max_f32(a, b + x1_);  \
    a = a + (x0);                           \
    b = nb_;                                \
    c = nc_;                                \
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
    const float x1 = lane_shr1(x0);
    const float x2 = lane_shr1(x1);
    const float nc = max_f32(b, c + x2);
    const float nb = (j <= L2 - 2) ? max_f32(a, b + x1) : NEG;  // one deletion must still follow
    if (j <= L2 - 3) a = a + x0;                                // two deletions must still follow
    b = nb;
    c = nc;
  }
#undef PSA_STEP
#undef PSA_T
  uint32_t key = 0u;
  if (lane >= 2 && s < d.s_end) {
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

// Table entries seen by Seq2 position j at offset s after 0, 1, 2 deletions.
// ---------------------------------------------------------------------------------
// passes 3 and 4 work on one offset that attains the maximum.  LDS holds the table
// entries Seq2 position j meets there after 0 deletions (d0[j]) and after 1 and 2
// deletions (d12[j] = {d1, d2}).  d12[L2-1].x is -inf: a chain with one deletion
// cannot consume the last position.  d12 is padded with 256 + 8 zero pairs so that
// lanes may run past the end: adding +0.0 changes nothing.
// ---------------------------------------------------------------------------------
__host__ __device__ inline int tie_d0_len(int len2) { return ((len2 + 8 + 3) / 4) * 4; }
__host__ __device__ inline int tie_d12_len(int len2) { return len2 + 256 + 8; }

__device__ __forceinline__ void fill_diagonals(float *d0, float2 *d12, const float *T, const ResolveParams &P,
                                               const SeqDesc &d, int s) {
  const int L2 = d.len2;
  for (int j = threadIdx.x; j < tie_d12_len(L2); j += blockDim.x) {
    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
    if (j < L2) {
      const int b = P.seq2_pool[d.off2 + j];
      const float *row = T + b * kRowStride;
      const int64_t i0 = (int64_t)s + j;
      const int c0 = (i0 < P.len1) ? P.seq1[i0] : kPadCode;
      const int c1 = (i0 - 1 >= 0 && i0 - 1 < P.len1) ? P.seq1[i0 - 1] : kPadCode;
      const int c2 = (i0 - 2 >= 0 && i0 - 2 < P.len1) ? P.seq1[i0 - 2] : kPadCode;
      v0 = row[c0];
      v1 = (j == L2 - 1) ? -INFINITY : row[c1];
      v2 = row[c2];
    }
    if (j < tie_d0_len(L2)) d0[j] = v0;
    d12[j] = make_float2(v1, v2);
  }
}

// Serial float32 chains over LDS arrays.  The adds are dependent by nature (the reference's
// rounding order); what can be hidden is the LDS latency: 8 values are in flight while
// the previous 8 are added, and the loop is unrolled twice so no register is copied.

// v += d0[from .. to), left to right (every lane computes the same value)
__device__ __forceinline__ float add_run_d0(const float *d0, int from, int to, float v) {
  int j = from;
  for (; j < to && (j & 3); ++j) v += d0[j];
  if (j + 8 <= to) {
    float4 a0 = *(const float4 *)(d0 + j), a1 = *(const float4 *)(d0 + j + 4);
    for (; j + 24 <= to; j += 16) {
      const float4 b0 = *(const float4 *)(d0 + j + 8), b1 = *(const float4 *)(d0 + j + 12);
      v += a0.x; v += a0.y; v += a0.z; v += a0.w; v += a1.x; v += a1.y; v += a1.z; v += a1.w;
      a0 = *(const float4 *)(d0 + j + 16);
      a1 = *(const float4 *)(d0 + j + 20);
      v += b0.x; v += b0.y; v += b0.z; v += b0.w; v += b1.x; v += b1.y; v += b1.z; v += b1.w;
    }
    v += a0.x; v += a0.y; v += a0.z; v += a0.w; v += a1.x; v += a1.y; v += a1.z; v += a1.w;
    j += 8;
  }
  for (; j < to; ++j) v += d0[j];
  return v;
}

// v += d12[from .. to).x, left to right (uniform)
__device__ __forceinline__ float add_run_d1(const float2 *d12, int from, int to, float v) {
  int j = from;
  for (; j < to && (j & 1); ++j) v += d12[j].x;
  if (j + 8 <= to) {
    const float4 *q = (const float4 *)(d12 + j);  // two positions per 16-byte read
    float4 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
    for (; j + 24 <= to; j += 16) {
      const float4 b0 = q[4], b1 = q[5], b2 = q[6], b3 = q[7];
      v += a0.x; v += a0.z; v += a1.x; v += a1.z; v += a2.x; v += a2.z; v += a3.x; v += a3.z;
      a0 = q[8];
      a1 = q[9];
      a2 = q[10];
      a3 = q[11];
      v += b0.x; v += b0.z; v += b1.x; v += b1.z; v += b2.x; v += b2.z; v += b3.x; v += b3.z;
      q += 8;
    }
    v += a0.x; v += a0.z; v += a1.x; v += a1.z; v += a2.x; v += a2.z; v += a3.x; v += a3.z;
    j += 8;
  }
  for (; j < to; ++j) v += d12[j].x;
  return v;
}

// pass 3: lane p computes max_q score(s,p,q) exactly: the chain that deleted p starts
// from a_p (the undeleted prefix sum) and then runs the same monotone recurrence,
//   c' = max(v, c + d2[j]),  v' = v + d1[j]   for j = p+1 .. L2-1.
// Lanes are skewed in time (lane p reads position p + t at step t), so every lane is in
// the same phase and no per-lane selects are needed.  The smallest p that attains the
// maximum wins.
__global__ __launch_bounds__(256) void psa_first_p(ResolveParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  __builtin_amdgcn_s_setprio(3);  // short dependent chains: win VALU arbitration against the enumeration waves
  __shared__ int skip_flag[1];
  float *T = (float *)smem;
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
    float *d0 = T + kDevTable;
    float2 *d12 = (float2 *)(d0 + tie_d0_len(L2));
    fill_diagonals(d0, d12, T, P, d, ao.s);
    __syncthreads();
    const float fstar = key2f(P.acc[ao.seq].res_key);
    const int p = p0 + tid;
    // a_p: prefix of the undeleted chain; lane p keeps the value before position p
    float v = add_run_d0(d0, 0, p0, 0.0f);
    float mine = v;
    const int pend = min(p0 + 256, L2 - 1);  // p <= L2 - 2
    for (int j = p0; j < pend; ++j) {
      mine = (j == p) ? v : mine;
      v += d0[j];
    }
    // skewed chain: step t handles position p + 1 + t
    const float2 *dp = d12 + p + 1;
    const int nsteps = L2 - 1 - p0;
    float c = -INFINITY;
    v = mine;
    int t = 0;
    if (nsteps >= 8) {
      float2 y[8], z[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) y[k] = dp[k];
      for (; t + 24 <= nsteps; t += 16) {
#pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = dp[t + 8 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          c = fmaxf(v, c + y[k].y);
          v = v + y[k].x;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = dp[t + 16 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          c = fmaxf(v, c + z[k].y);
          v = v + z[k].x;
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        c = fmaxf(v, c + y[k].y);
        v = v + y[k].x;
      }
      t += 8;
    }
    for (; t < nsteps; ++t) {
      const float2 x = dp[t];
      c = fmaxf(v, c + x.y);
      v = v + x.x;
    }
    int cand = (p <= L2 - 2 && c == fstar) ? p : INT_MAX;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) cand = min(cand, __shfl_xor(cand, m));
    if ((tid & 63) == 0 && cand != INT_MAX) atomicMin(&P.acc[ao.seq].res_p, cand);
  }
}

// pass 4: with p* fixed, lane q evaluates score(s,p*,q) exactly as the reference adds
// it: the one-deletion chain value before position q, then d2[j] for j > q (skewed in
// time like pass 3).  The smallest q, then the smallest offset, wins.
__global__ __launch_bounds__(256) void psa_first_q(ResolveParams P) {
  extern __shared__ __align__(16) unsigned char smem[];
  __builtin_amdgcn_s_setprio(3);  // short dependent chains: win VALU arbitration against the enumeration waves
  __shared__ int skip_flag[1];
  float *T = (float *)smem;
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
    float *d0 = T + kDevTable;
    float2 *d12 = (float2 *)(d0 + tie_d0_len(L2));
    fill_diagonals(d0, d12, T, P, d, ao.s);
    __syncthreads();
    const float fstar = key2f(P.acc[ao.seq].res_key);
    const int q = q0 + tid;
    float v = add_run_d0(d0, 0, pstar, 0.0f);  // a_{p*}
    int j = pstar + 1;
    if (j < q0) {  // positions every lane of the block consumes
      v = add_run_d1(d12, j, q0, v);
      j = q0;
    }
    float mine = v;
    const int qend = min(q0 + 256, L2);
    for (; j < qend; ++j) {
      mine = (j == q) ? v : mine;               // value handed over when q is deleted
      v += d12[j].x;
    }
    const float2 *dp = d12 + q + 1;
    const int nsteps = L2 - 1 - q0;
    float c = mine;
    int t = 0;
    if (nsteps >= 8) {
      float y[8], z[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) y[k] = dp[k].y;
      for (; t + 24 <= nsteps; t += 16) {
#pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = dp[t + 8 + k].y;
#pragma unroll
        for (int k = 0; k < 8; ++k) c += y[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = dp[t + 16 + k].y;
#pragma unroll
        for (int k = 0; k < 8; ++k) c += z[k];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) c += y[k];
      t += 8;
    }
    for (; t < nsteps; ++t) c += dp[t].y;
    unsigned long long cand = ~0ull;
    if (q > pstar && q <= L2 - 1 && c == fstar) cand = ((unsigned long long)(uint32_t)q << 32) | (uint32_t)ao.s;
    cand = wave_min_u64(cand);
    if ((tid & 63) == 0 && cand != ~0ull) atomicMin(&P.acc[ao.seq].res_loc, cand);
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
  __shared__ __align__(4) unsigned char lds_class[kDevTable];
  const int i = blockIdx.x;
  static_assert(kDevTable % 4 == 0 && kDevTable / 4 <= 256, "one 4-byte load per thread");
  if (threadIdx.x < kDevTable / 4) ((uint32_t *)lds_class)[threadIdx.x] = ((const uint32_t *)P.class_t)[threadIdx.x];
  const SeqDesc d = P.desc[i];
  const SeqAcc a = P.acc[i];
  const SeqAcc ea = P.eacc[i];
  psa_result *r = P.out + i;
  const int tid = threadIdx.x;
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

hipError_t launch_init(SeqAcc *acc, int n, unsigned *counters, int ncounters, hipStream_t st) {
  int m = n > ncounters ? n : ncounters;
  hipLaunchKernelGGL(psa_init_acc, dim3((m + 255) / 256), dim3(256), 0, st, acc, n, counters, ncounters);
  return hipGetLastError();
}

static size_t fwd_lds(int max_len2) {
  return (size_t)kDevTable * 4 + 4 * (size_t)fwd_rows(max_len2) + 4 * (size_t)fwd_window(max_len2) + 64;
}
static size_t tie_lds(int max_len2) {
  return (size_t)kDevTable * 4 + 4 * (size_t)tie_d0_len(max_len2) + 8 * (size_t)tie_d12_len(max_len2);
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

hipError_t launch_finalize(const FinalParams &P, int n, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
  hipExtLaunchKernelGGL(psa_finalize, dim3(n), dim3(256), 0, st, ev_start, ev_stop, 0, P);
  return hipGetLastError();
}


}  // namespace psa
