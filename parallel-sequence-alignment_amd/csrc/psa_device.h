// Device-side helpers shared by the kernel files.
#ifndef PSA_DEVICE_H
#define PSA_DEVICE_H

#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>

#include "psa_kernels.h"

namespace psa {

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
  c.fg_row = (uint32_t)uni((int)p->fg_row);
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
  d.enum_ns = uni(p->enum_ns);
  d.enum_slack = uni(p->enum_slack);
  d.pad_ = 0;
  return d;
}

}  // namespace psa

#endif
