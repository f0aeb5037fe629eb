// psa_score_range_noalloc: the float32-exact answer for one Seq2 over a range of offsets, for
// callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
// that forbid hipMalloc).  Inputs and the output record are device memory the caller owns; every
// intermediate the kernels need -- residue codes, descriptor, forward blocks, accumulators,
// per-offset maxima, the list of offsets that attain the maximum, the transposed table and the
// class table -- lives in one fixed array in device memory that is part of the library's code
// object.  Nothing here calls hipMalloc, hipFree, a stream or event constructor, or any routine
// that allocates; the kernels are the ones psa_score_range runs (psa_fwd, psa_collect,
// psa_first_p, psa_first_q, psa_finalize), given pointers into that array.
// The reference has no counterpart: its calc_best_score_CUDA allocates five device buffers per
// call and never frees them (P/cudaFunctions.cu:273-288).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "psa_device.h"

namespace psa {
namespace {

constexpr int64_t kNaMaxLen1 = PSA_NOALLOC_MAX_SEQ1;  // characters of Seq1 (and, plus 3, offsets)
constexpr int64_t kNaMaxOffsets = kNaMaxLen1 + 3;
constexpr size_t na_align(size_t x) { return (x + 255) / 256 * 256; }

// byte offsets inside the array
constexpr size_t kOffDesc = 0;                                              // SeqDesc
constexpr size_t kOffAcc = kOffDesc + na_align(sizeof(SeqDesc));            // 2 x SeqAcc (resolve, unused enumeration copy)
constexpr size_t kOffCnt = kOffAcc + na_align(2 * sizeof(SeqAcc));          // [0] active offsets, [1] invalid-residue flag
constexpr size_t kOffTable = kOffCnt + 256;                                 // float [27][32], [seq2 code][seq1 code]
constexpr size_t kOffClass = kOffTable + na_align(kDevTable * sizeof(float));  // class digits, same layout
constexpr size_t kOffSeq2 = kOffClass + na_align(kDevTable);                // residue codes of Seq2
constexpr size_t kOffSeq1 = kOffSeq2 + na_align(PSA_MAX_SEQ2 + 16);         // residue codes of Seq1
constexpr size_t kOffFwd = kOffSeq1 + na_align((size_t)kNaMaxLen1);         // forward blocks
constexpr size_t kNaMaxFwd = (size_t)((kNaMaxOffsets + kFwdOffsets - 1) / kFwdOffsets);
constexpr size_t kOffFs = kOffFwd + na_align(kNaMaxFwd * sizeof(FwdBlock));  // per-offset maxima
constexpr size_t kOffActive = kOffFs + na_align((size_t)kNaMaxOffsets * sizeof(float));
constexpr size_t kScratchBytes = kOffActive + na_align((size_t)kNaMaxOffsets * sizeof(ActiveOffset));
static_assert(kScratchBytes == PSA_NOALLOC_SCRATCH_BYTES, "psa.h states the size of the array");

}  // namespace
}  // namespace psa

// The array.  Each GPU the library runs on holds its own copy, at its own address, from the
// moment the code object is loaded there until the process ends.  In host code this name is
// only a handle for hipGetSymbolAddress, never an address.
static __device__ __align__(256) unsigned char scratch[psa::kScratchBytes];

namespace psa {
namespace {

struct NaSetup {
  const char *seq1;  // device, 'A'..'Z'
  const char *seq2;  // device, 'A'..'Z'
  const float *table;  // device, 26 x 26, row = Seq1 residue (the argument of calc_best_score_CUDA)
  unsigned char *base;  // the array
  psa_result *out;      // device-accessible
  int64_t len1;
  int32_t len2;
  int32_t s_begin, s_end;
  int32_t nfwd;
  unsigned char cls[kDevTable];  // class digits [seq2 code][seq1 code]
};

// Everything the resolve kernels expect to find in device memory, built on the device: residue
// codes (a character outside 'A'..'Z' becomes the padding code and raises the flag), descriptor,
// forward blocks, accumulators and counters in their start state, the transposed table, the
// class table, and the host-known fields of the record.
__global__ __launch_bounds__(256) void psa_na_setup(NaSetup P) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
  unsigned *cnt = (unsigned *)(P.base + kOffCnt);
  unsigned char *c1 = P.base + kOffSeq1, *c2 = P.base + kOffSeq2;
  bool bad = false;
  for (int64_t i = gid; i < P.len1; i += gsz) {
    const unsigned ch = (unsigned char)P.seq1[i];
    const bool ok = ch >= 'A' && ch <= 'Z';
    bad |= !ok;
    c1[i] = ok ? (unsigned char)(ch - 'A') : (unsigned char)kPadCode;
  }
  for (int64_t i = gid; i < P.len2 + 16; i += gsz) {
    unsigned char code = (unsigned char)kPadCode;
    if (i < P.len2) {
      const unsigned ch = (unsigned char)P.seq2[i];
      const bool ok = ch >= 'A' && ch <= 'Z';
      bad |= !ok;
      if (ok) code = (unsigned char)(ch - 'A');
    }
    c2[i] = code;
  }
  if (bad) atomicOr(&cnt[1], 1u);
  float *tt = (float *)(P.base + kOffTable);
  unsigned char *ct = P.base + kOffClass;
  for (int64_t i = gid; i < kDevTable; i += gsz) {
    const int b = (int)(i / kRowStride), a = (int)(i % kRowStride);
    const float v = (a < 26 && b < 26) ? P.table[a * 26 + b] : 0.0f;
    if (!isfinite(v)) atomicOr(&cnt[1], 2u);  // the kernels' maxima are built on finite entries (as psa_set_table checks)
    tt[i] = v;
    ct[i] = P.cls[i];
  }
  FwdBlock *fwd = (FwdBlock *)(P.base + kOffFwd);
  for (int64_t i = gid; i < P.nfwd; i += gsz) {
    FwdBlock fb;
    fb.seq = 0;
    fb.s0 = P.s_begin + (int32_t)i * kFwdOffsets;
    fwd[i] = fb;
  }
  if (gid == 0) {
    SeqDesc d;
    d.fs_base = 0;
    d.off2 = 0;
    d.len2 = P.len2;
    d.s_begin = P.s_begin;
    d.s_end = P.s_end;
    d.frac_bits = -1;
    d.enum_ns = 1;
    d.enum_slack = -1;
    d.pad_ = 0;
    *(SeqDesc *)(P.base + kOffDesc) = d;
    SeqAcc a;
    a.enum_key = 0ull;
    a.enum_loc = ~0ull;
    a.res_loc = ~0ull;
    a.res_key = 0u;
    a.res_p = INT_MAX;
    a.short_best = 0ull;
    a.short_s = ~0u;
    a.short_count = 0u;
    SeqAcc *acc = (SeqAcc *)(P.base + kOffAcc);
    acc[0] = a;
    acc[1] = a;
    cnt[0] = 0u;
    psa_result r;
    memset(&r, 0, sizeof r);
    r.enum_frac_bits = -1;
    r.enum_exact = -1;
    r.status = PSA_OK;
    r.candidates = (int64_t)(P.s_end - P.s_begin) * ((int64_t)P.len2 * (P.len2 - 1) / 2);
    *P.out = r;
  }
}

// cnt[1], the invalid-residue flag, is cleared on the caller's stream before psa_na_setup (whose blocks
// set it in no particular order, so it cannot clear it itself): a call that failed half-way, or was
// abandoned, leaves nothing behind for the next one.
__global__ void psa_na_epilogue(unsigned char *base, psa_result *out) {
  unsigned *cnt = (unsigned *)(base + kOffCnt);
  if (cnt[1]) {
    psa_result r;
    memset(&r, 0, sizeof r);
    r.enum_frac_bits = -1;
    r.enum_exact = -1;
    r.status = (cnt[1] & 1u) ? PSA_ERR_RESIDUE : PSA_ERR_ARG;  // bit 0: a character outside 'A'..'Z'; bit 1: a table entry not finite
    *out = r;
  }
}

constexpr int kNaMaxDevices = 64;
std::mutex g_na_mu;
unsigned char *g_na_ptr[kNaMaxDevices] = {};  // the array's address on each device, once asked for

thread_local std::string g_na_err;

// Once per device, with that device current: the array's address there, and the dynamic-LDS
// limits of the resolve kernels on that device.
int na_device_setup(int device, unsigned char **out) {
  if (device < 0 || device >= kNaMaxDevices) {
    g_na_err = "psa_score_range_noalloc: device index beyond the table of per-device addresses";
    return PSA_ERR_ARG;
  }
  std::lock_guard<std::mutex> lock(g_na_mu);
  if (!g_na_ptr[device]) {
    unsigned char *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
      g_na_err = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
      return PSA_ERR_HIP;
    }
    if ((e = set_resolve_kernel_attributes()) != hipSuccess) {
      g_na_err = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e);
      return PSA_ERR_HIP;
    }
    g_na_ptr[device] = ptr;
  }
  *out = g_na_ptr[device];
  return PSA_OK;
}

}  // namespace
}  // namespace psa

extern "C" {

const char *psa_noalloc_last_error_text(void) { return psa::g_na_err.c_str(); }

int psa_noalloc_init(void) {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) {
    psa::g_na_err = "no HIP device visible";
    return PSA_ERR_NO_DEVICE;
  }
  unsigned char *base = nullptr;
  return psa::na_device_setup(device, &base);
}

int psa_score_range_noalloc(const char *d_seq1, int64_t len1, const char *d_seq2, int32_t len2,
                            const float *d_table, int64_t s_begin, int64_t s_end, void *stream, psa_result *d_out) {
  using namespace psa;
  g_na_err.clear();  // the text belongs to the most recent call on this thread
  if (!d_seq1 || !d_seq2 || !d_table || !d_out || len1 < 1) {
    g_na_err = "psa_score_range_noalloc: null pointer or empty Seq1";
    return PSA_ERR_ARG;
  }
  if (len1 > kNaMaxLen1) {
    g_na_err = "psa_score_range_noalloc: Seq1 is longer than PSA_NOALLOC_MAX_SEQ1 characters, the fixed capacity of "
               "this entry point (use psa_score_range, which allocates what it needs)";
    return PSA_ERR_LIMIT;
  }
  if (len2 < 2) {
    g_na_err = "psa_score_range_noalloc: Seq2 shorter than 2 characters";
    return PSA_ERR_SEQ2_SHORT;
  }
  if (len2 > PSA_MAX_SEQ2 || (int64_t)len2 - 2 > len1) {
    g_na_err = "psa_score_range_noalloc: Seq2 longer than Seq1 + 2 or than PSA_MAX_SEQ2";
    return PSA_ERR_SEQ2_LONG;
  }
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) {
    g_na_err = "no HIP device visible";
    return PSA_ERR_NO_DEVICE;
  }
  unsigned char *base = nullptr;  // the array on this device: this pointer, never the symbol, from here on
  int st = na_device_setup(device, &base);
  if (st != PSA_OK) return st;
  hipStream_t st_ = (hipStream_t)stream;
  const int64_t S = len1 - len2 + 3;  // P/cudaFunctions.cu:269
  const int64_t sb = s_begin < 0 ? 0 : s_begin, se = s_end > S ? S : s_end;
  const bool empty = se <= sb;  // a shard without an offset: a record without a candidate

  NaSetup A{};
  A.seq1 = d_seq1;
  A.seq2 = d_seq2;
  A.table = d_table;
  A.base = base;
  A.out = d_out;
  A.len1 = len1;
  A.len2 = len2;
  A.s_begin = (int32_t)(empty ? 0 : sb);
  A.s_end = (int32_t)(empty ? 0 : se);
  A.nfwd = empty ? 0 : (int32_t)((se - sb + kFwdOffsets - 1) / kFwdOffsets);
  if ((size_t)A.nfwd > kNaMaxFwd) {  // cannot happen: len1 is within the capacity
    g_na_err = "psa_score_range_noalloc: more forward blocks than the fixed capacity";
    return PSA_ERR_LIMIT;
  }
  for (int b = 0; b < 27; ++b)
    for (int a = 0; a < kRowStride; ++a)
      A.cls[b * kRowStride + a] = (a < 26 && b < 26) ? (unsigned char)residue_class_code(a, b) : (unsigned char)3;
  const int setup_blocks = (int)((len1 + 255) / 256 > 1024 ? 1024 : (len1 + 255) / 256);
  // (a memset node on the caller's stream: allocates nothing, and may be captured into a graph)
  hipError_t e = hipMemsetAsync(base + kOffCnt + sizeof(unsigned), 0, sizeof(unsigned), st_);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(psa_na_setup, dim3(setup_blocks), dim3(256), 0, st_, A);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = debug_sync("psa_na_setup");
  if (e == hipSuccess && !empty) {
    ResolveParams P{};
    P.seq1 = base + kOffSeq1;
    P.len1 = len1;
    P.seq2_pool = base + kOffSeq2;
    P.desc = (const SeqDesc *)(base + kOffDesc);
    P.fwd_blocks = (const FwdBlock *)(base + kOffFwd);
    P.acc = (SeqAcc *)(base + kOffAcc);
    P.table_t = (const float *)(base + kOffTable);
    P.fs = (float *)(base + kOffFs);
    P.active = (ActiveOffset *)(base + kOffActive);
    P.nactive = (unsigned *)(base + kOffCnt);
    P.pchunks = (len2 + 255) / 256;
    P.one_seq = 1;
    e = launch_resolve(P, A.nfwd, len2, 512, st_, nullptr);
  }
  if (e == hipSuccess) {
    FinalParams F{};
    F.seq1 = base + kOffSeq1;
    F.seq2_pool = base + kOffSeq2;
    F.desc = (const SeqDesc *)(base + kOffDesc);
    F.acc = (SeqAcc *)(base + kOffAcc);
    F.eacc = (SeqAcc *)(base + kOffAcc) + 1;
    F.nactive = (unsigned *)(base + kOffCnt);
    F.class_t = base + kOffClass;
    F.out = d_out;
    F.stages = PSA_STAGE_RESOLVE;
    F.which = PSA_STAGE_RESOLVE | PSA_STAGE_ENUMERATE;  // the enumeration fields are cleared
    e = launch_finalize(F, 1, st_, nullptr, nullptr);
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(psa_na_epilogue, dim3(1), dim3(1), 0, st_, base, d_out);
    e = hipGetLastError();
  }
  if (e != hipSuccess) {
    g_na_err = std::string("psa_score_range_noalloc: ") + hipGetErrorString(e);
    return PSA_ERR_HIP;
  }
  return PSA_OK;
}

}  // extern "C"
