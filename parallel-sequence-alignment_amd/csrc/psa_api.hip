// C ABI of libpsa_hip: contexts, batches, launch orchestration and the drop-in shim.
// Citations: P/ = "Parallel Solution/", S/ = "Sequential Solution/".
#include <hip/hip_runtime.h>

#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "psa_kernels.h"

namespace {

thread_local std::string g_err;

#define PSA_HIP(call)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      g_err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
      return PSA_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)

constexpr size_t kSlabUnit = 256;                 // bytes: one run of 64 int32
constexpr size_t kSlabMinUnits = (5u << 20) / 256;  // the largest chunk (64 + 1 rows of 128 + 1 runs) fits
constexpr size_t kSlabDefaultMiB = 256;  // against 512: one long pair 1.8 % faster, a list of 4096 Seq2 0.4 %, half the memory (2048: 3.4 % slower on the pair)

// Device block, pinned host block and events of one batch.  A context keeps the bundles of
// destroyed batches and hands them to new ones, so that scoring batch after batch through
// psa_score_batch (or Seq2 after Seq2 through the drop-in shim) allocates nothing after the
// first call; a bundle that is too small is grown.
struct Bundle {
  unsigned char *dev = nullptr;
  size_t dev_cap = 0;
  unsigned char *pin = nullptr;      // pinned and mapped: upload staging, then the result records
  unsigned char *pin_dev = nullptr;  // the device's address of pin
  size_t pin_cap = 0;
  hipEvent_t ev[8] = {};
  std::vector<hipEvent_t> pass_ev;  // 2 per slab pass and slab
};

void bundle_free(Bundle *u) {
  if (!u) return;
  if (u->dev) (void)hipFree(u->dev);
  if (u->pin) (void)hipHostFree(u->pin);
  for (hipEvent_t e : u->ev)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : u->pass_ev)
    if (e) (void)hipEventDestroy(e);
  delete u;
}

constexpr size_t kMaxCachedBundles = 4;
constexpr size_t kSmallUpload = 64u << 10;  // work lists up to this size are fetched by the first run's set-up kernel
constexpr int kMaxSlabs = 3;

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

struct psa_ctx {
  int device = 0;
  int num_cus = 256;
  // Four streams, all of one priority.  Candidate loops alternate between st_enum[0] and
  // st_enum[1]: consecutive ones share nothing (each has its slab and its copy of the
  // accumulators), so the next one fills the CUs that the draining one leaves idle; locate and
  // the enumeration half of the records follow their candidate loop on its stream.  st_res
  // carries the float32 resolve chain, st_pre the pre-passes and the uploads of new batches.
  hipStream_t st_enum[2] = {nullptr, nullptr}, st_res = nullptr, st_pre = nullptr;
  int enum_streams = 2, enum_flip = 0;
  // Arithmetic of the enumeration stage: float32 holding integers of at most 24 bits (default;
  // the chip issues float add + max3 faster than the integer pair) or int32 with 30-bit values
  // (PSA_ENUM_I32=1, 6 more fractional bits in the fixed-point table).
  bool enum_f32 = true;
  double table_abs_max = 0.0;  // largest |entry| of the current table
  // Slabs of g rows.  Three: with two, the candidate loops of consecutive passes start as soon as
  // their pre-passes are done, share the GPU, end together, and both slabs wait for their
  // pre-passes while nothing else can run (a tenth of the time on many-pass batches); the third
  // slab is filled while a pair of loops runs, so a loop is always ready when one ends.
  int nslabs = kMaxSlabs;
  int *d_slab[kMaxSlabs] = {};
  size_t slab_units = 0;  // capacity of each slab in runs of 64 int32
  int slab_count = 0;
  hipEvent_t ev_prep[kMaxSlabs] = {};
  // Bound on the work in flight: every kFencePasses-th slab pass records a fence behind its candidate loop, and
  // before a fence event is used again the host waits for its previous record, so at most kFences x
  // kFencePasses passes (a few thousand launches) of the context are ever enqueued and not yet finished.
  // Without it a caller that enqueues two runs of a 4096-Seq2 list back to back has some six thousand
  // launches and their completion signals outstanding -- measured on ROCm 7.2: from then on, and until that
  // process ends, every OTHER process that initialises HIP on the machine dies in the runtime's event thread.
  static constexpr int kFences = 8, kFencePasses = 64;
  hipEvent_t fence[kFences] = {};
  bool fence_used[kFences] = {};
  int fence_at = 0, passes_since_fence = 0;
  hipEvent_t slab_reader[kMaxSlabs] = {};  // end of the last candidate loop that read slab i (a batch's event)
  psa_batch *slab_reader_batch[kMaxSlabs] = {};
  int slab_flip = 0;
  unsigned char *d_seq1 = nullptr;
  int64_t len1 = 0;
  float table[PSA_TABLE_SIZE];
  psa::FracScan frac_scan;  // of the table: the fixed-point scale per Seq2 follows from it and the length
  bool have_table = false, have_seq1 = false;
  float *d_table_t = nullptr;          // [27][32] floats, [seq2 code][seq1 code]
  unsigned char *d_class_t = nullptr;  // [27][32] class digits
  std::vector<Bundle *> spare;         // bundles of destroyed batches
  unsigned long long *d_probe = nullptr;  // clock probe of the candidate loop: cycles, 100 MHz ticks, units
  bool probe_on = false;
  bool shortlist = true;  // enumerating runs shortlist and replay the candidates near the maximum (PSA_SHORTLIST=0: off)
  // Batches created on this context and not yet destroyed.  A batch keeps values derived from
  // Seq1 and the table (offset ranges, forward blocks, the fixed-point scale), so neither may be
  // replaced while one exists; psa_shutdown detaches the ones a caller left behind.
  std::vector<psa_batch *> live;
};

struct psa_batch {
  psa_ctx *ctx = nullptr;
  int n = 0;
  int max_len2 = 0;
  int64_t total_offsets = 0;
  int nvalid = 0;
  std::vector<psa::SeqDesc> desc;
  std::vector<psa_result> tmpl;  // per-Seq2 record with the host-known fields filled in
  unsigned char *d_pool = nullptr;
  psa::SeqDesc *d_desc = nullptr;
  psa::SeqAcc *d_acc = nullptr;
  unsigned *d_counters = nullptr;  // [0] = the active-offset count, [1], [2] = shortlist rows of the two runs in flight, [3], [4] = their record blocks done
  // enumeration stage: chunks sorted by falling Seq2 length; a slab pass covers chunks
  // [c0, c1), whose rows fit one slab
  struct SlabPass {
    int c0 = 0, c1 = 0;
    int buf = 0;  // slab of the context this pass uses
    int cls_c0[psa::kEnumClasses + 1] = {};  // chunks of kernel class k: [cls_c0[k + 1], cls_c0[k]) (classes fall)
    psa::WaveParams wave{};
    // start and end of the candidate loop, one pair per slab (a one-pass batch alternates slabs)
    hipEvent_t ev_b[kMaxSlabs] = {}, ev_e[kMaxSlabs] = {};
    int last_buf = 0;
  };
  std::vector<SlabPass> passes;
  bool two_slabs = false;
  psa::Chunk *d_chunks = nullptr;
  unsigned long long *d_chunk_best = nullptr;
  int nchunks = 0;
  int cls_c0[psa::kEnumClasses + 1] = {};  // the same over the whole chunk list
  psa::FwdBlock *d_fwd = nullptr;
  int nfwd = 0;
  float *d_fs = nullptr;
  psa::ActiveOffset *d_active = nullptr;
  psa_result *h_out = nullptr;  // pinned and mapped: psa_finalize writes the records straight into it
  psa_result *d_out = nullptr;  // the device's address of h_out
  // shortlists of the enumeration stage (psa_internal.h): short_cap entries per Seq2, two copies like the
  // accumulators; what the record kernel reports of them lands beside the records
  psa::ShortRow *d_short = nullptr;
  int short_cap = 0;  // rows of one copy of the list
  int short_quota = psa::kShortRowsPerSeq;  // rows one Seq2 may list
  psa::ShortOut *h_short = nullptr, *d_short_out = nullptr;
  bool pending_short = false;  // the enqueued run fills h_short
  // start of the first / end of the last kernel on the enumeration side and on the resolve side
  hipEvent_t ev_first = nullptr, ev_last = nullptr, ev_res_first = nullptr, ev_res = nullptr, ev_init = nullptr;
  hipEvent_t ev_upload = nullptr;  // the work lists have arrived on the device
  // The enumeration accumulators (SeqAcc::enum_*, the per-chunk keys) exist twice and runs
  // alternate between the copies, so that the candidate loop of a run can start while locate
  // and the records of the run before are still reading theirs.
  int acc_par = 0;
  bool cap_loops = true;  // the candidate loops leave room on every CU for the short kernels (small batches)
  bool mfma_loop = true;  // the candidate loop runs on the matrix cores (float32 arithmetic, the default)
  hipEvent_t ev_tail[2] = {nullptr, nullptr};  // copy i has been read and reset: end of that run's record kernel
  bool tail_recorded[2] = {false, false};
  int tail_stream[2] = {0, 0};                 // the candidate-loop stream that event is on
  // a run enqueued in two parts (one shard of a long pair: the ranks exchange their enumerated maxima between
  // the candidate loops and locate): what the first part leaves for the second
  int head_es = 0;                                    // stream of the last candidate loop
  hipEvent_t head_last_loop[2] = {nullptr, nullptr};  // end of the last candidate loop on each loop stream
  bool global_ref = false;  // the pending run shortlisted against a maximum given from outside (see psa_batch_wait)
  bool last_run_res_only = false;  // the run before wrote both halves of the records on st_res
  bool res_recorded = false;       // ev_res has been recorded (some run had a record kernel on st_res)
  hipEvent_t enum_only_tail = nullptr;  // record kernel of a run without the resolve stage, not yet ordered before st_res
  float enum_ms = 0.f, total_ms = 0.f;
  bool ran = false;
  bool pending = false, pending_timed = false, pending_resolve = false;  // a run is enqueued, not yet waited for
  bool pending_both = false;  // ... with both stages: its records get the cross-check of the two maxima
  bool acc_ready = false;  // the accumulators were reset after the previous run
  Bundle *res = nullptr;   // owns every device and pinned pointer above and the events
  bool upload_in_flight = false;  // the asynchronous upload of the work lists has not been waited for
  // small batches: the work lists stay in the pinned block and the first run's set-up kernel copies them
  const unsigned char *setup_src = nullptr;
  size_t setup_bytes = 0;
};

extern "C" {

const char *psa_last_error_text(void) { return g_err.c_str(); }

int psa_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int psa_init(int device, psa_ctx **out) {
  if (!out) return PSA_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    g_err = "no HIP device visible";
    return PSA_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) return PSA_ERR_ARG;
  PSA_HIP(hipSetDevice(device));
  psa_ctx *c = new (std::nothrow) psa_ctx();
  if (!c) return PSA_ERR_NOMEM;
  // (a failure below releases what the context already holds)
#define PSA_HIP_C(call)                                          \
  do {                                                           \
    hipError_t e_ = (call);                                      \
    if (e_ != hipSuccess) {                                      \
      g_err = std::string(#call) + ": " + hipGetErrorString(e_); \
      psa_shutdown(c);                                           \
      return PSA_ERR_HIP;                                        \
    }                                                            \
  } while (0)
  c->device = device;
  hipDeviceProp_t prop;
  PSA_HIP_C(hipGetDeviceProperties(&prop, device));
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char *e = std::getenv("PSA_ENUM_I32")) c->enum_f32 = std::atoi(e) == 0;
  if (const char *e = std::getenv("PSA_SHORTLIST")) c->shortlist = std::atoi(e) != 0;
  // dynamic-LDS limits of the kernels: an attribute of (kernel, device), so set per context
  PSA_HIP_C(psa::set_enum_kernel_attributes());
  PSA_HIP_C(psa::set_resolve_kernel_attributes());
  // Every stream has the same priority.  (With the short kernels' streams at a higher priority
  // than the candidate loop's, a stream that waited for a candidate loop not yet dispatched hung
  // the run: measured.)  The short kernels find room beside the candidate loops because those
  // cap themselves at three workgroups per CU (see enqueue_run), not through priority.  Four
  // streams, because a process has four hardware queues here: a fifth stream shares a queue
  // with another one and runs in order with it.
  for (hipStream_t &st : c->st_enum) PSA_HIP_C(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  PSA_HIP_C(hipStreamCreateWithFlags(&c->st_res, hipStreamNonBlocking));
  PSA_HIP_C(hipStreamCreateWithFlags(&c->st_pre, hipStreamNonBlocking));
  if (const char *e = std::getenv("PSA_ENUM_STREAMS")) c->enum_streams = std::atoi(e) == 1 ? 1 : 2;  // tuning aid
  if (const char *e = std::getenv("PSA_SLABS")) c->nslabs = std::atoi(e) == 2 ? 2 : kMaxSlabs;          // tuning aid
  for (hipEvent_t &e : c->ev_prep) PSA_HIP_C(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (hipEvent_t &e : c->fence) PSA_HIP_C(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  PSA_HIP_C(hipMalloc((void **)&c->d_table_t, psa::kDevTable * sizeof(float)));
  PSA_HIP_C(hipMalloc((void **)&c->d_class_t, psa::kDevTable));
  PSA_HIP_C(hipMalloc((void **)&c->d_probe, 64 * 8 * sizeof(unsigned long long)));
  PSA_HIP_C(hipMemset(c->d_probe, 0, 64 * 8 * sizeof(unsigned long long)));
  std::vector<unsigned char> cls(psa::kDevTable, 3);
  for (int b = 0; b < 26; ++b)
    for (int a = 0; a < 26; ++a) cls[b * psa::kRowStride + a] = (unsigned char)psa::residue_class_code(a, b);
  PSA_HIP_C(hipMemcpy(c->d_class_t, cls.data(), cls.size(), hipMemcpyHostToDevice));
#undef PSA_HIP_C
  *out = c;
  return PSA_OK;
}

void psa_shutdown(psa_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  // Batches the caller has not destroyed: their runs are awaited, their buffers released, and the
  // handles stay valid for psa_batch_destroy only (every other call on them returns PSA_ERR_STATE).
  if (!c->live.empty()) {
    (void)hipDeviceSynchronize();
    for (psa_batch *b : c->live) {
      bundle_free(b->res);
      b->res = nullptr;
      b->ctx = nullptr;
      b->pending = b->ran = false;
    }
    c->live.clear();
  }
  if (c->d_seq1) (void)hipFree(c->d_seq1);
  if (c->d_table_t) (void)hipFree(c->d_table_t);
  if (c->d_class_t) (void)hipFree(c->d_class_t);
  if (c->d_probe) (void)hipFree(c->d_probe);
  for (hipStream_t st : c->st_enum)
    if (st) (void)hipStreamDestroy(st);
  if (c->st_res) (void)hipStreamDestroy(c->st_res);
  if (c->st_pre) (void)hipStreamDestroy(c->st_pre);
  for (int *p : c->d_slab)
    if (p) (void)hipFree(p);
  for (Bundle *u : c->spare) bundle_free(u);
  for (hipEvent_t e : c->ev_prep)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->fence)
    if (e) (void)hipEventDestroy(e);
  delete c;
}

int psa_set_seq1(psa_ctx *c, const char *seq1, int64_t len1) {
  if (!c || !seq1) return PSA_ERR_ARG;
  if (len1 < 0) len1 = (int64_t)std::strlen(seq1);
  if (len1 < 1 || len1 > 0x7fff0000ll) return PSA_ERR_ARG;
  if (!c->live.empty()) {
    g_err = "psa_set_seq1: the context still has batches (they were laid out for the current Seq1)";
    return PSA_ERR_STATE;
  }
  std::vector<unsigned char> codes((size_t)len1);
  for (int64_t i = 0; i < len1; ++i) {
    unsigned char ch = (unsigned char)seq1[i];
    if (ch < 'A' || ch > 'Z') return PSA_ERR_RESIDUE;
    codes[(size_t)i] = ch - 'A';
  }
  PSA_HIP(hipSetDevice(c->device));
  PSA_HIP(hipDeviceSynchronize());  // runs enqueued and not yet waited for still read the old Seq1
  if (c->d_seq1) {
    PSA_HIP(hipFree(c->d_seq1));
    c->d_seq1 = nullptr;
  }
  PSA_HIP(hipMalloc((void **)&c->d_seq1, (size_t)len1));
  PSA_HIP(hipMemcpy(c->d_seq1, codes.data(), (size_t)len1, hipMemcpyHostToDevice));
  c->len1 = len1;
  c->have_seq1 = true;
  return PSA_OK;
}

int psa_set_table(psa_ctx *c, const float table[PSA_TABLE_SIZE]) {
  if (!c || !table) return PSA_ERR_ARG;
  if (!c->live.empty()) {
    g_err = "psa_set_table: the context still has batches (their fixed-point scale follows the current table)";
    return PSA_ERR_STATE;
  }
  for (int i = 0; i < PSA_TABLE_SIZE; ++i)
    if (!std::isfinite(table[i])) {
      g_err = "psa_set_table: entry " + std::to_string(i) + " is not finite";
      return PSA_ERR_ARG;
    }
  std::memcpy(c->table, table, sizeof c->table);
  c->table_abs_max = 0.0;
  for (float x : c->table) c->table_abs_max = std::max(c->table_abs_max, (double)std::fabs(x));
  psa::frac_table_scan(c->table, &c->frac_scan);
  // device copy: transposed to [seq2 code][seq1 code], zero row and column for the padding code
  std::vector<float> t(psa::kDevTable, 0.0f);
  for (int a = 0; a < 26; ++a)
    for (int b = 0; b < 26; ++b) t[b * psa::kRowStride + a] = table[a * 26 + b];
  PSA_HIP(hipSetDevice(c->device));
  PSA_HIP(hipDeviceSynchronize());  // runs enqueued and not yet waited for still read the old table
  PSA_HIP(hipMemcpy(c->d_table_t, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
  c->have_table = true;
  return PSA_OK;
}

int psa_clock_probe(psa_ctx *c, int enable, double *clock_mhz, double *wave_cycles_per_unit) {
  if (!c) return PSA_ERR_ARG;
  PSA_HIP(hipSetDevice(c->device));
  PSA_HIP(hipDeviceSynchronize());
  unsigned long long slots[64 * 8], v[3] = {0, 0, 0};
  PSA_HIP(hipMemcpy(slots, c->d_probe, sizeof slots, hipMemcpyDeviceToHost));
  PSA_HIP(hipMemset(c->d_probe, 0, sizeof slots));
  for (int i = 0; i < 64; ++i)
    for (int k = 0; k < 3; ++k) v[k] += slots[8 * i + k];
  if (clock_mhz) *clock_mhz = v[1] ? 100.0 * (double)v[0] / (double)v[1] : 0.0;
  if (wave_cycles_per_unit) *wave_cycles_per_unit = v[2] ? (double)v[0] / (double)v[2] : 0.0;
  c->probe_on = enable != 0;
  return PSA_OK;
}

int psa_set_weights(psa_ctx *c, const float weights[4]) {
  if (!c || !weights) return PSA_ERR_ARG;
  float t[PSA_TABLE_SIZE];
  psa::build_table_from_weights(weights, t);
  return psa_set_table(c, t);
}

void psa_batch_destroy(psa_batch *b) {
  if (!b) return;
  if (b->ctx) {
    std::vector<psa_batch *> &live = b->ctx->live;
    live.erase(std::remove(live.begin(), live.end(), b), live.end());
    (void)hipSetDevice(b->ctx->device);
    if (b->pending) (void)hipDeviceSynchronize();  // a run nobody waited for
    for (int i = 0; i < kMaxSlabs; ++i)
      if (b->ctx->slab_reader_batch[i] == b) {
        (void)hipEventSynchronize(b->ctx->slab_reader[i]);
        b->ctx->slab_reader[i] = nullptr;
        b->ctx->slab_reader_batch[i] = nullptr;
      }
  }
  if (b->res) {
    if (b->ctx && b->upload_in_flight) (void)hipStreamSynchronize(b->ctx->st_pre);
    if (b->ctx && b->ctx->spare.size() < kMaxCachedBundles)
      b->ctx->spare.push_back(b->res);
    else
      bundle_free(b->res);
  }
  delete b;
}

// stages: the stages the batch will ever run, if the caller knows (the one-shot entry points do);
// a batch that never enumerates gets no chunk list, no per-chunk keys and no slabs
static int batch_build(psa_ctx *c, int n, const char *const *seq2, const int32_t *len2, bool ranged, int64_t rb,
                       int64_t re, uint32_t stages, psa_batch **out) {
  if (!out) return PSA_ERR_ARG;
  *out = nullptr;
  if (!c || n < 0 || (n > 0 && !seq2)) return PSA_ERR_ARG;
  if (!c->have_seq1 || !c->have_table) return PSA_ERR_STATE;
  PSA_HIP(hipSetDevice(c->device));
  psa_batch *b = new (std::nothrow) psa_batch();
  if (!b) return PSA_ERR_NOMEM;
  b->ctx = c;
  c->live.push_back(b);
  b->n = n;
  b->desc.resize((size_t)n);
  b->tmpl.assign((size_t)n, psa_result{});
  std::vector<unsigned char> pool;
  std::vector<psa::Chunk> chunks;
  std::vector<psa::FwdBlock> fwd;
  int64_t fs_total = 0;

  for (int i = 0; i < n; ++i) {
    psa::SeqDesc &d = b->desc[(size_t)i];
    psa_result &r = b->tmpl[(size_t)i];
    d = psa::SeqDesc{};
    d.frac_bits = -1;
    r.enum_frac_bits = -1;
    r.enum_exact = -1;
    const char *s2 = seq2[i];
    if (!s2) {
      r.status = PSA_ERR_ARG;
      continue;
    }
    int64_t L2 = len2 ? len2[i] : (int64_t)std::strlen(s2);
    if (L2 < 2) {
      r.status = PSA_ERR_SEQ2_SHORT;
      continue;
    }
    if (L2 - 2 > c->len1 || L2 > PSA_MAX_SEQ2) {
      r.status = PSA_ERR_SEQ2_LONG;
      continue;
    }
    bool ok = true;
    for (int64_t j = 0; j < L2 && ok; ++j) ok = s2[j] >= 'A' && s2[j] <= 'Z';
    if (!ok) {
      r.status = PSA_ERR_RESIDUE;
      continue;
    }
    const int64_t S = c->len1 - L2 + 3;
    int64_t sb = 0, se = S;
    if (ranged) {
      sb = std::max<int64_t>(rb, 0);
      se = std::min<int64_t>(re, S);
      if (se <= sb) {
        // a shard that owns no offset (more shards than offsets): a valid record without a
        // candidate (n = 0, candidates = 0), which psa_combine_results passes over
        r.status = PSA_OK;
        continue;
      }
    }
    d.off2 = (int32_t)pool.size();
    for (int64_t j = 0; j < L2; ++j) pool.push_back((unsigned char)(s2[j] - 'A'));
    while (pool.size() % 16) pool.push_back((unsigned char)psa::kPadCode);
    d.len2 = (int32_t)L2;
    d.s_begin = (int32_t)sb;
    d.s_end = (int32_t)se;
    d.fs_base = fs_total;
    fs_total += se - sb;
    int exact = -1;
    d.frac_bits = psa::choose_frac_bits(c->frac_scan, L2, c->enum_f32 ? psa::kSumBitsF32 : psa::kSumBitsI32, &exact);
    if (psa::enum_class_of((int)L2) < 0) d.frac_bits = -1;
    r.enum_frac_bits = d.frac_bits;
    r.enum_exact = d.frac_bits >= 0 ? exact : -1;
    d.enum_slack = -1;
    if (d.frac_bits >= 0 && c->shortlist && (stages & PSA_STAGE_ENUMERATE)) {
      // Window of the shortlist: for one candidate of m table entries |fl(sum) - E 2^-r| <= bnd, so the
      // candidate with the highest float32 sum has E >= (enumerated maximum) - 2 bnd 2^r.  The float32 term
      // vanishes when every partial sum is exact (level 2), the table term when the table is (level >= 1).
      const double m = (double)(L2 - 2), u = std::ldexp(1.0, -24);
      double bnd = 0.0;
      if (exact < 2 && m > 1.0) bnd += (m - 1.0) * u / (1.0 - (m - 1.0) * u) * m * c->table_abs_max;
      if (exact < 1) bnd += m * std::ldexp(1.0, -(d.frac_bits + 1));
      const double units = 2.0 * std::ldexp(bnd, d.frac_bits);
      if (units < (double)(1 << 28)) d.enum_slack = bnd > 0.0 ? (int32_t)units + 1 : 0;
    }
    r.candidates = (se - sb) * (L2 * (L2 - 1) / 2);
    r.status = PSA_OK;
    b->max_len2 = std::max<int>(b->max_len2, (int)L2);
    b->nvalid++;
    for (int64_t s0 = sb; s0 < se; s0 += psa::kFwdOffsets) fwd.push_back(psa::FwdBlock{i, (int32_t)s0});
  }
  b->total_offsets = fs_total;

  // Enumeration chunks, longest Seq2 first.  A wave of psa_enum_wave scores enum_ns
  // offsets x 256 positions p; a chunk (the pre-pass unit) is 2^sub_log2 such sub-runs.
  std::vector<int> order;
  for (int i = 0; i < n && (stages & PSA_STAGE_ENUMERATE); ++i)
    if (b->tmpl[(size_t)i].status == PSA_OK && b->desc[(size_t)i].frac_bits >= 0) order.push_back(i);
  std::stable_sort(order.begin(), order.end(),
                   [&](int x, int y) { return b->desc[(size_t)x].len2 > b->desc[(size_t)y].len2; });
  constexpr int RW = psa::kEnumWaveRuns;
  auto runs_of = [](int len2) { return (len2 + 63) / 64; };
  auto levels_of = [&](int len2) { return (runs_of(len2) + RW - 1) / RW; };
  int64_t row_units = 0;
  for (int i : order) {
    const psa::SeqDesc &d = b->desc[(size_t)i];
    row_units += (int64_t)(d.s_end - d.s_begin) * levels_of(d.len2);
  }
  // enough units to keep every wave slot busy many times over, each unit long enough
  // (about 30 us: 72000 cycles at ~333 cycles per 64 x 64 candidates) to hide its start-up;
  // a batch too small for that still gets two to three rounds of units over the wave slots (for
  // Seq1 = 3000, 4 x Seq2 = 2000: three offsets per unit; two or four are 4 to 7 % slower).
  // The matrix-core candidate loop runs 4 waves per SIMD, the vector-ALU one 8.
  const bool mfma_loop = c->enum_f32 && !(std::getenv("PSA_ENUM_MFMA") && std::atoi(std::getenv("PSA_ENUM_MFMA")) == 0);
  b->mfma_loop = mfma_loop;
  const int64_t slots = (int64_t)c->num_cus * (mfma_loop ? 16 : 32);
  const int64_t ns_small = mfma_loop ? std::min<int64_t>(4, (2 * row_units) / (5 * slots))
                                     : std::min<int64_t>(2, (row_units + slots) / (2 * slots));
  const int64_t ns_cap = std::max<int64_t>({1, row_units / (16 * slots), ns_small});
  // A batch with many rounds of units per wave slot lets its candidate loops take every wave
  // slot (8 waves per SIMD: 7 to 10 % faster than 6); the short kernels of the other streams then
  // move in as workgroups retire, which delays them by a few unit times -- nothing against such
  // a run.  A small batch keeps the cap: its resolve chain is as long as its candidate loop.
  b->cap_loops = row_units < 16 * slots;
  int min_ns = psa::kEnumMaxNs;
  for (int i : order) {
    psa::SeqDesc &d = b->desc[(size_t)i];
    // cycles of the longest level of one offset: 64 x 64 blocks at ~333 cycles on the vector ALU,
    // or 32 x 32 tiles at ~46 on the matrix cores (8 per block of 32 q beyond the level's own 256, 36 inside)
    const int64_t unit_cycles = mfma_loop ? ((int64_t)std::max(0, 2 * runs_of(d.len2) - 8) * 8 + 36) * 46
                                          : (int64_t)std::max(1, runs_of(d.len2) - RW) * RW * 333;
    // (the matrix-core loop gains 2 to 3 % from units of about half a millisecond: one long pair of
    // 8000 x 200000 runs 16 offsets per unit at 4.92e13 candidates/s against 4.76e13 with one)
    const int64_t target_cycles = mfma_loop ? 1000000 : 72000;
    int64_t ns = std::min<int64_t>(std::max<int64_t>(1, (target_cycles + unit_cycles / 2) / unit_cycles), ns_cap);
    // (the matrix-core loop: at most 24 offsets per unit -- a list of 512 Seq2 of 500 to 2000 characters runs
    // 1.5 % faster than with 64, whose units of a millisecond pack badly into slab passes of two)
    if (mfma_loop) ns = std::min<int64_t>(ns, 24);
    if (const char *e = std::getenv("PSA_ENUM_NS")) ns = std::atoi(e);  // tuning aid
    d.enum_ns = (int32_t)std::max<int64_t>(1, std::min<int64_t>(ns, psa::kEnumMaxNs));
    min_ns = std::min(min_ns, d.enum_ns);
  }
  // chunks of at least 4 offsets: a chunk's pre-pass scans two diagonals and stores one row more than it has
  // offsets, but locate walks a chunk it visits offset by offset, and on a small batch it sits between one
  // candidate loop and the next but one (chunks of at least 8: one-shot calls 5 % slower on 4 x Seq2 = 2000,
  // steps with a host wait 16 % slower on the reference's input.txt)
  int sub_log2 = min_ns >= 4 ? 0 : (min_ns >= 2 ? 1 : 2);
  if (const char *e = std::getenv("PSA_ENUM_SUB_LOG2")) sub_log2 = std::max(0, std::min(3, std::atoi(e)));  // tuning aid
  size_t total_units = 0;
  for (int i : order) {
    const psa::SeqDesc &d = b->desc[(size_t)i];
    const int64_t per = std::min<int64_t>((int64_t)d.enum_ns << sub_log2, psa::kEnumMaxNs);
    for (int64_t s0 = d.s_begin; s0 < d.s_end; s0 += per) {
      const int64_t cnt = std::min<int64_t>(per, d.s_end - s0);
      chunks.push_back(psa::Chunk{i, (int32_t)s0, (int32_t)cnt, 0u});
      total_units += ((size_t)cnt + 1) * ((size_t)runs_of(d.len2) + 1);  // one row more than offsets, see psa_enum_prep
    }
  }
  // slab passes: consecutive chunks whose rows fit one slab
  size_t cap_units = kSlabDefaultMiB * ((1u << 20) / kSlabUnit);
  if (const char *e = std::getenv("PSA_SLAB_MB")) cap_units = (size_t)std::max(1, std::atoi(e)) * ((1u << 20) / kSlabUnit);
  cap_units = std::max(kSlabMinUnits, std::min(cap_units, total_units));
  {
    size_t used = 0;
    for (size_t ci = 0; ci < chunks.size(); ++ci) {
      psa::Chunk &ch = chunks[ci];
      const size_t need = ((size_t)ch.ns + 1) * ((size_t)runs_of(b->desc[(size_t)ch.seq].len2) + 1);
      if (b->passes.empty() || used + need > cap_units) {
        if (!b->passes.empty()) b->passes.back().c1 = (int)ci;
        b->passes.emplace_back();
        b->passes.back().c0 = (int)ci;
        used = 0;
      }
      ch.fg_row = (uint32_t)used;
      used += need;
    }
    if (!b->passes.empty()) b->passes.back().c1 = (int)chunks.size();
    bool two_slabs = b->passes.size() > 1;
    for (size_t k = 0; k < b->passes.size(); ++k) {
      b->passes[k].buf = (int)(k % (size_t)c->nslabs);
    }
    b->two_slabs = two_slabs;
  }
  // A small batch runs all its chunks through the pre-pass and locate variant of its longest
  // Seq2: one launch each instead of one per variant present (a variant works for any shorter
  // Seq2; on a few thousand chunks the idle threads cost less than the extra launches).
  const bool one_class = chunks.size() <= 2048;
  const int top_class = psa::enum_class_of(b->max_len2);
  auto class_bounds = [&](int c0, int c1, int *out) {
    // chunks [c0, c1) are sorted by falling length, so kernel classes fall too:
    // class k owns [out[k + 1], out[k])
    int pos = c0;
    out[psa::kEnumClasses] = c0;
    for (int k = psa::kEnumClasses - 1; k >= 0; --k) {
      while (pos < c1 &&
             (one_class ? top_class : psa::enum_class_of(b->desc[(size_t)chunks[(size_t)pos].seq].len2)) == k)
        ++pos;
      out[k] = pos;
    }
  };
  class_bounds(0, (int)chunks.size(), b->cls_c0);
  // PSA_ENUM_BAND=B (tuning aid): B adjacent levels of a sub-run side by side on one XCD (WaveParams).  Measured on
  // one Seq2 of 8000 against a Seq1 of 200000: four times fewer bytes fetched into the L2s, and 3 to 10 % fewer
  // candidates per second for every B from 2 to 32 -- level-major numbering stays the default.
  int env_band = 0;
  if (const char *e = std::getenv("PSA_ENUM_BAND")) env_band = std::max(0, std::atoi(e));
  for (psa_batch::SlabPass &sp : b->passes) {
    class_bounds(sp.c0, sp.c1, sp.cls_c0);
    psa::WaveParams &W = sp.wave;
    W.sub_log2 = sub_log2;
    int64_t base = 0;
    int prefix = sp.c1 - sp.c0;  // chunks that have level w: a prefix of the pass, shrinking as w grows
    for (int w = 0; w <= psa::kEnumMaxLevels; ++w) W.level_base[w] = INT32_MAX;
    W.level_base[0] = 0;
    for (int w = 0; w < psa::kEnumMaxLevels; ++w) {
      while (prefix > 0 && levels_of(b->desc[(size_t)chunks[(size_t)(sp.c0 + prefix - 1)].seq].len2) <= w) --prefix;
      if (prefix == 0) break;
      W.level_base[w] = (int)std::min<int64_t>(base, INT32_MAX);
      base += (int64_t)prefix << sub_log2;
    }
    if (base > INT32_MAX) {
      g_err = "too many enumeration units in one slab pass";
      psa_batch_destroy(b);
      return PSA_ERR_ARG;
    }
    W.nunits = (int)base;
    int band = env_band;
    W.nlev = sp.c1 > sp.c0 ? levels_of(b->desc[(size_t)chunks[(size_t)sp.c0].seq].len2) : 1;
    band = std::min(band, W.nlev);
    W.band = band;
    W.ncols = (sp.c1 - sp.c0) << sub_log2;
    W.ncols8 = (W.ncols + 7) / 8 * 8;
    if (band > 0) {
      const int64_t nu = (int64_t)W.ncols8 * ((W.nlev + band - 1) / band * band);
      if (nu > INT32_MAX) {
        g_err = "too many enumeration units in one slab pass";
        psa_batch_destroy(b);
        return PSA_ERR_ARG;
      }
      W.nunits = (int)nu;
    }
  }

  auto fail = [&](int code) {
    psa_batch_destroy(b);
    return code;
  };
  b->nfwd = (int)fwd.size();
  b->nchunks = (int)chunks.size();
  if (b->nchunks > 0) {
    // slabs of the context: grown when this batch needs more, two once a batch needs two passes
    const int want = c->nslabs;
    if (cap_units > c->slab_units || want > c->slab_count) {
      const size_t units = std::max(cap_units, c->slab_units);
      const int count = std::max(want, c->slab_count);
      (void)hipDeviceSynchronize();  // runs enqueued on other batches may still use the old slabs
      for (int i = 0; i < kMaxSlabs; ++i) {
        if (c->d_slab[i]) (void)hipFree(c->d_slab[i]);
        c->d_slab[i] = nullptr;
      }
      c->slab_units = 0;
      c->slab_count = 0;
      for (int i = 0; i < count; ++i) {
        // one extra run: the candidate loop's per-lane load of g[p + 1] reads one value past the last row
        if (hipMalloc((void **)&c->d_slab[i], (units + 1) * kSlabUnit) != hipSuccess) {
          g_err = "hipMalloc(enumeration slab) failed";
          return fail(PSA_ERR_HIP);
        }
        if (hipMemset(c->d_slab[i], 0, (units + 1) * kSlabUnit) != hipSuccess) {
          g_err = "hipMemset(enumeration slab) failed";
          return fail(PSA_ERR_HIP);
        }
      }
      c->slab_units = units;
      c->slab_count = count;
    }
  }
#define PSA_HIP_B(call)                                               \
  do {                                                                \
    hipError_t e_ = (call);                                           \
    if (e_ != hipSuccess) {                                           \
      g_err = std::string(#call) + ": " + hipGetErrorString(e_);      \
      return fail(PSA_ERR_HIP);                                       \
    }                                                                 \
  } while (0)
  // One device block and one pinned block per batch.  Uploaded part first (Seq2 pool,
  // descriptors, forward blocks, chunks), then what only the kernels touch.
  const size_t o_pool = 0;
  const size_t o_desc = align_up(o_pool + pool.size(), 256);
  const size_t o_fwd = align_up(o_desc + b->desc.size() * sizeof(psa::SeqDesc), 256);
  const size_t o_chunks = align_up(o_fwd + fwd.size() * sizeof(psa::FwdBlock), 256);
  const size_t upload_bytes = align_up(o_chunks + chunks.size() * sizeof(psa::Chunk), 256);
  const size_t o_best = upload_bytes;
  const size_t o_acc = align_up(o_best + 2 * chunks.size() * sizeof(unsigned long long), 256);
  const size_t o_cnt = align_up(o_acc + 2 * (size_t)n * sizeof(psa::SeqAcc), 256);
  const size_t o_fs = align_up(o_cnt + 64, 256);
  const size_t o_active = align_up(o_fs + (size_t)fs_total * sizeof(float), 256);
  // The shortlist of a run has room for every Seq2's full quota of rows, so that a record depends on nothing
  // but its own Seq2 (a list split over ranks returns the records of the whole list, run after run).  The
  // quota is kShortRowsPerSeq unless the list would pass kShortBudgetRows (more than 16384 Seq2 in one batch).
  int short_cap = 0, short_quota = psa::kShortRowsPerSeq;
  if (c->shortlist && !chunks.empty()) {
    if (const char *e = std::getenv("PSA_SHORT_ROWS")) short_quota = std::max(1, std::min(psa::kShortRowsPerSeq, std::atoi(e)));  // test aid
    const int64_t nv = std::max(1, b->nvalid);
    if (nv * short_quota > psa::kShortBudgetRows) short_quota = (int)std::max<int64_t>(1, psa::kShortBudgetRows / nv);
    short_cap = (int)(nv * short_quota);
  }
  b->short_quota = short_quota;
  b->short_cap = short_cap;
  const size_t o_short = align_up(o_active + (size_t)fs_total * sizeof(psa::ActiveOffset), 256);
  const size_t dev_bytes = align_up(o_short + 2 * (size_t)short_cap * sizeof(psa::ShortRow), 256) + 256;
  const size_t o_out = upload_bytes;  // in the pinned block the records follow the staging area
  const size_t o_sout = align_up(o_out + std::max<size_t>(1, (size_t)n) * sizeof(psa_result), 64);
  const size_t pin_bytes = o_sout + std::max<size_t>(1, (size_t)n) * sizeof(psa::ShortOut);
  {
    Bundle *u = nullptr;
    for (size_t i = 0; i < c->spare.size(); ++i)
      if (c->spare[i]->dev_cap >= dev_bytes && c->spare[i]->pin_cap >= pin_bytes) {
        u = c->spare[i];
        c->spare.erase(c->spare.begin() + (long)i);
        break;
      }
    if (!u && !c->spare.empty()) {  // grow the most recent one instead of piling up small ones
      u = c->spare.back();
      c->spare.pop_back();
    }
    if (!u) u = new (std::nothrow) Bundle();
    if (!u) return fail(PSA_ERR_NOMEM);
    b->res = u;
    if (u->dev_cap < dev_bytes) {
      if (u->dev) (void)hipFree(u->dev);
      u->dev = nullptr;
      u->dev_cap = 0;
      const size_t cap = dev_bytes + dev_bytes / 4;
      PSA_HIP_B(hipMalloc((void **)&u->dev, cap));
      u->dev_cap = cap;
    }
    if (u->pin_cap < pin_bytes) {
      if (u->pin) (void)hipHostFree(u->pin);
      u->pin = nullptr;
      u->pin_cap = 0;
      const size_t cap = pin_bytes + pin_bytes / 4;
      PSA_HIP_B(hipHostMalloc((void **)&u->pin, cap, hipHostMallocMapped));
      u->pin_cap = cap;
      PSA_HIP_B(hipHostGetDevicePointer((void **)&u->pin_dev, u->pin, 0));
    }
    for (hipEvent_t &e : u->ev)
      if (!e) PSA_HIP_B(hipEventCreate(&e));
    while (u->pass_ev.size() < 2 * kMaxSlabs * b->passes.size()) {
      hipEvent_t e = nullptr;
      PSA_HIP_B(hipEventCreate(&e));
      u->pass_ev.push_back(e);
    }
    // stage the work lists in pinned memory and send them with one asynchronous copy on the
    // pre-pass stream (beside whatever runs are enqueued); the first run waits for its event
    if (!pool.empty()) std::memcpy(u->pin + o_pool, pool.data(), pool.size());
    if (!b->desc.empty()) std::memcpy(u->pin + o_desc, b->desc.data(), b->desc.size() * sizeof(psa::SeqDesc));
    if (!fwd.empty()) std::memcpy(u->pin + o_fwd, fwd.data(), fwd.size() * sizeof(psa::FwdBlock));
    if (!chunks.empty()) std::memcpy(u->pin + o_chunks, chunks.data(), chunks.size() * sizeof(psa::Chunk));
    if (upload_bytes <= kSmallUpload) {
      // a few KB: the first run's set-up kernel reads them through the mapping of the pinned block
      // (one launch on the resolve stream instead of a copy, an event, a memset and a kernel)
      b->setup_src = u->pin_dev;
      b->setup_bytes = upload_bytes;
    } else {
      PSA_HIP_B(hipMemcpyAsync(u->dev, u->pin, upload_bytes, hipMemcpyHostToDevice, c->st_pre));
      PSA_HIP_B(hipEventRecord(u->ev[5], c->st_pre));
      b->ev_upload = u->ev[5];
      b->upload_in_flight = true;
    }
    b->d_pool = u->dev + o_pool;
    b->d_desc = (psa::SeqDesc *)(u->dev + o_desc);
    b->d_fwd = (psa::FwdBlock *)(u->dev + o_fwd);
    b->d_chunks = (psa::Chunk *)(u->dev + o_chunks);
    b->d_chunk_best = (unsigned long long *)(u->dev + o_best);
    b->d_acc = (psa::SeqAcc *)(u->dev + o_acc);
    b->d_counters = (unsigned *)(u->dev + o_cnt);
    b->d_fs = (float *)(u->dev + o_fs);
    b->d_active = (psa::ActiveOffset *)(u->dev + o_active);
    b->h_out = (psa_result *)(u->pin + o_out);
    if (n > 0) std::memcpy(b->h_out, b->tmpl.data(), (size_t)n * sizeof(psa_result));
    b->d_out = (psa_result *)(u->pin_dev + o_out);
    b->d_short = short_cap > 0 ? (psa::ShortRow *)(u->dev + o_short) : nullptr;
    b->h_short = (psa::ShortOut *)(u->pin + o_sout);
    b->d_short_out = (psa::ShortOut *)(u->pin_dev + o_sout);
    if (n > 0) std::memset(b->h_short, 0, (size_t)n * sizeof(psa::ShortOut));
    b->ev_first = u->ev[0];
    b->ev_last = u->ev[1];
    b->ev_init = u->ev[2];
    b->ev_res = u->ev[3];
    b->ev_res_first = u->ev[4];
    b->ev_tail[0] = u->ev[6];
    b->ev_tail[1] = u->ev[7];
    for (size_t k = 0; k < b->passes.size(); ++k)
      for (int i = 0; i < kMaxSlabs; ++i) {
        b->passes[k].ev_b[i] = u->pass_ev[2 * kMaxSlabs * k + (size_t)i];
        b->passes[k].ev_e[i] = u->pass_ev[2 * kMaxSlabs * k + kMaxSlabs + (size_t)i];
      }
  }
#undef PSA_HIP_B
  *out = b;
  return PSA_OK;
}

int psa_batch_create(psa_ctx *c, int n, const char *const *seq2, const int32_t *len2, psa_batch **out) {
  return batch_build(c, n, seq2, len2, false, 0, 0, PSA_STAGE_ENUMERATE | PSA_STAGE_RESOLVE, out);
}

int psa_batch_create_range(psa_ctx *c, const char *seq2, int32_t len2, int64_t s_begin, int64_t s_end,
                           psa_batch **out) {
  const char *one[1] = {seq2};
  int32_t l[1] = {len2};
  return batch_build(c, 1, one, len2 >= 0 ? l : nullptr, true, s_begin, s_end, PSA_STAGE_ENUMERATE | PSA_STAGE_RESOLVE, out);
}

int psa_batch_size(const psa_batch *b) { return b ? b->n : 0; }

int psa_batch_passes(const psa_batch *b) { return b ? (int)b->passes.size() : 0; }

// Enqueues one run on the streams of the context: pre-passes on st_pre; candidate loops
// alternating between the two st_enum, the last one followed on its stream by locate and the
// enumeration half of the records; the resolve chain with its half of the records on st_res.
// No host synchronisation.  The events that time the run ride on kernel dispatches, the host
// waits for the record kernels, and the accumulators are left reset by their last readers.
// A run can be enqueued in two parts, with the host in between: kPartHead = set-up, pre-passes, candidate
// loops, and the resolve chain with its half of the records; kPartTail = locate, the shortlist replay and
// the enumeration half of the records.  One shard of a long pair runs that way: its rank waits for the
// candidate loops, exchanges the enumerated maximum with the other ranks, and the second part shortlists
// against the maximum over all shards (*short_ref).  A whole run is both parts at once, in the order
// loops, locate and records, resolve chain.
constexpr unsigned kPartHead = 1u, kPartTail = 2u;
static int enqueue_run(psa_batch *b, uint32_t stages, bool *timed, unsigned parts = kPartHead | kPartTail,
                       const int *short_ref = nullptr) {
  psa_ctx *c = b->ctx;
  const bool resolve = (stages & PSA_STAGE_RESOLVE) && b->nfwd > 0;
  const bool enumerate = (stages & PSA_STAGE_ENUMERATE) && b->nchunks > 0;
  const bool head = (parts & kPartHead) != 0, tail = (parts & kPartTail) != 0;
  // only a batch's first run, or one after a failed run, needs the set-up kernel
  const bool init = head && !b->acc_ready;
  // The first run of a small one-pass batch (every one-shot call) is a chain set-up -> pre-pass -> candidate loop
  // with nothing beside it: all three go on the candidate loop's stream, in order.  On three streams the two
  // cross-stream waits cost about 20 us each, a sixth of such a call.
  static const bool no_fast_start = std::getenv("PSA_NO_FAST_START") != nullptr;  // measurement aid
  const bool fast_start = init && enumerate && b->passes.size() == 1 && b->setup_src && !no_fast_start;
  hipStream_t st_fast = fast_start ? c->st_enum[c->enum_streams > 1 ? (c->enum_flip ^ 1) : 0] : nullptr;  // (the pass below takes it)
  if (init) {
    // on st_res, which never waits for a candidate loop
    if (b->setup_src) {
      PSA_HIP(psa::launch_setup_small(b->setup_src, b->d_pool, b->setup_bytes, b->d_acc, 2 * b->n, b->d_counters, 5,
                                      b->d_chunk_best, 2 * (size_t)b->nchunks, fast_start ? st_fast : c->st_res));
      PSA_HIP(psa::debug_sync("psa_setup_small"));
    } else {
      if (b->upload_in_flight) PSA_HIP(hipStreamWaitEvent(c->st_res, b->ev_upload, 0));
      PSA_HIP(psa::launch_init(b->d_acc, 2 * b->n, b->d_counters, 5, c->st_res));
      PSA_HIP(psa::debug_sync("psa_init_acc"));
      if (b->nchunks > 0)
        PSA_HIP(hipMemsetAsync(b->d_chunk_best, 0, 2 * (size_t)b->nchunks * sizeof(unsigned long long), c->st_res));
    }
    PSA_HIP(hipEventRecord(b->ev_init, fast_start ? st_fast : c->st_res));
    // (a fast start puts the other streams behind the set-up once its candidate loop is on its way: nothing
    // of this batch reaches them before that)
    if (!fast_start) {
      PSA_HIP(hipStreamWaitEvent(c->st_pre, b->ev_init, 0));
      for (hipStream_t st : c->st_enum) PSA_HIP(hipStreamWaitEvent(st, b->ev_init, 0));
    }
    b->tail_recorded[0] = b->tail_recorded[1] = false;
  }
  if (head) b->acc_ready = false;
  *timed = false;
  if (enumerate && head) b->acc_par ^= 1;
  const int par = b->acc_par;
  psa::SeqAcc *const eacc = b->d_acc + (size_t)par * (size_t)b->n;
  unsigned long long *const chunk_best = b->d_chunk_best + (size_t)par * (size_t)b->nchunks;
  psa::FinalParams F{};
  F.seq1 = c->d_seq1;
  F.seq2_pool = b->d_pool;
  F.desc = b->d_desc;
  F.acc = b->d_acc;
  F.eacc = eacc;
  F.nactive = b->d_counters;
  F.class_t = c->d_class_t;
  F.out = b->d_out;
  F.stages = stages;
  psa::EnumParams P{};
  if (enumerate) {
    P.seq1 = c->d_seq1;
    P.len1 = c->len1;
    P.seq2_pool = b->d_pool;
    P.desc = b->d_desc;
    P.acc = eacc;
    P.table_t = c->d_table_t;
    P.short_rows = b->d_short ? b->d_short + (size_t)par * (size_t)b->short_cap : nullptr;
    P.short_total = b->d_short ? b->d_counters + 1 + par : nullptr;
    P.short_rows_cap = b->short_cap;
    P.short_quota = b->short_quota;
    F.short_quota = b->short_quota;
    P.prep_wave_form = b->cap_loops ? 0 : 1;  // (large batches: see psa_enum_prep_wave)
    P.has_short_ref = short_ref ? 1 : 0;
    P.short_ref = short_ref ? *short_ref : 0;
    F.short_total = P.short_total;
    F.short_out = b->d_short_out;
    F.short_empty_ok = short_ref ? 1 : 0;
  }
  if (enumerate && head) {
    // A pre-pass may fill a slab as soon as the last candidate loop that read it has finished,
    // so it runs beside the candidate loop of the pass before it (on the other slab) and, for
    // runs enqueued back to back, beside the candidate loop of the run before.
    const bool piped = b->passes.size() > 1;
    hipEvent_t ev_first = b->ev_first;              // start of the run's first enumeration kernel
    hipEvent_t last_loop[2] = {nullptr, nullptr};  // per candidate-loop stream
    int es = 0;
    for (size_t k = 0; k < b->passes.size(); ++k) {
      psa_batch::SlabPass &sp = b->passes[k];
      // a one-pass batch alternates between the slabs from run to run, so that the pre-pass of a
      // run enqueued behind another runs beside that one's candidate loop
      const int buf = piped ? sp.buf : (c->slab_flip = (c->slab_flip + 1) % c->nslabs);
      hipStream_t st_prep = fast_start ? st_fast : c->st_pre;
      if (c->slab_reader[buf]) PSA_HIP(hipStreamWaitEvent(st_prep, c->slab_reader[buf], 0));
      int last_cls = -1;
      for (int cls = 0; cls < psa::kEnumClasses; ++cls)
        if (sp.cls_c0[cls] > sp.cls_c0[cls + 1]) {
          last_cls = cls;
          break;
        }
      for (int cls = psa::kEnumClasses - 1; cls >= 0; --cls) {
        const int c0 = sp.cls_c0[cls + 1], c1 = sp.cls_c0[cls];
        if (c1 <= c0) continue;
        P.chunks = b->d_chunks + c0;
        P.chunk_best = chunk_best + c0;
        P.nchunks = c1 - c0;
        P.fg = c->d_slab[buf];
        PSA_HIP(psa::launch_enum_prep(cls, P, c->enum_f32, st_prep, ev_first,
                                      cls == last_cls ? c->ev_prep[buf] : nullptr));
        ev_first = nullptr;
      }
      es = c->enum_streams > 1 ? (c->enum_flip ^= 1) : 0;
      hipStream_t st_loop = c->st_enum[es];
      // this run's copy of the accumulators was last used two runs ago, maybe on the other stream
      if (b->tail_recorded[par] && !last_loop[es] && b->tail_stream[par] != es)
        PSA_HIP(hipStreamWaitEvent(st_loop, b->ev_tail[par], 0));
      if (st_loop != st_prep) PSA_HIP(hipStreamWaitEvent(st_loop, c->ev_prep[buf], 0));
      psa::WaveParams W = sp.wave;
      W.chunks = b->d_chunks + sp.c0;
      W.chunk_best = chunk_best + sp.c0;
      W.desc = b->d_desc;
      W.acc = eacc;
      W.clock_probe = c->probe_on ? c->d_probe : nullptr;
      // 41 KB of LDS per workgroup (untouched) keep the candidate loops, two of which may be in
      // flight, to three workgroups of 8 waves per CU between them: the short kernels of the
      // other streams always find wave slots, registers and LDS
      const bool mfma_loop = b->mfma_loop;
      int lds = b->cap_loops && !mfma_loop ? 41 * 1024 : 0;
      if (const char *e = std::getenv("PSA_ENUM_LDS_KB")) lds = std::max(0, std::min(64, std::atoi(e))) * 1024;  // tuning aid
      // small batches (capped launches) use workgroups of 8 waves, the others workgroups of one wave,
      // which take a wave slot as soon as one is free.  (Small batches with the matrix-core loop: whole
      // steps of 4 x Seq2 = 2000 enqueued back to back take 0.180 ms with 8-wave workgroups, 0.192 ms
      // with 4 and 0.25 ms with 1 or 2: the short kernels of the other streams are placed sooner
      // when a retiring workgroup frees two wave slots per SIMD at once.)
      int wpg = b->cap_loops ? 8 : 1;
      if (const char *e = std::getenv("PSA_ENUM_WPG")) wpg = std::max(1, std::min(8, std::atoi(e)));  // tuning aid
      static const bool skip_loop = std::getenv("PSA_SKIP_LOOP") != nullptr;  // measurement aid: the pre-pass alone (records are void)
      if (skip_loop) W.nunits = std::min(W.nunits, 1);
      PSA_HIP(psa::launch_enum_wave(c->d_slab[buf], W, c->enum_f32, mfma_loop, wpg, lds, st_loop, sp.ev_b[buf], sp.ev_e[buf]));
      if (++c->passes_since_fence >= psa_ctx::kFencePasses) {
        const int slot = c->fence_at;
        if (c->fence_used[slot]) PSA_HIP(hipEventSynchronize(c->fence[slot]));  // the host runs too far ahead
        PSA_HIP(hipEventRecord(c->fence[slot], st_loop));
        c->fence_used[slot] = true;
        c->fence_at = (slot + 1) % psa_ctx::kFences;
        c->passes_since_fence = 0;
      }
      sp.last_buf = buf;
      c->slab_reader[buf] = sp.ev_e[buf];
      c->slab_reader_batch[buf] = b;
      last_loop[es] = sp.ev_e[buf];
    }
    b->head_es = es;
    b->head_last_loop[0] = last_loop[0];
    b->head_last_loop[1] = last_loop[1];
  }
  if (fast_start) {
    PSA_HIP(hipStreamWaitEvent(c->st_res, b->ev_init, 0));
    PSA_HIP(hipStreamWaitEvent(c->st_pre, b->ev_init, 0));
    for (hipStream_t st : c->st_enum)
      if (st != st_fast) PSA_HIP(hipStreamWaitEvent(st, b->ev_init, 0));
  }
  // locate, the shortlist replay and the enumeration half of the records; behind the resolve chain in the host's
  // order of enqueueing when the run is a fast start (the candidate loop gives the host time for both, and the
  // resolve chain then runs beside the loop's first half instead of beside locate)
  auto enqueue_tail = [&]() -> int {
    // locate and the records follow the last candidate loop on its stream; the candidate loops
    // of the next run start on the other stream, on the other copy of the accumulators
    const int es = b->head_es;
    hipStream_t st_tail = c->st_enum[es];
    if (b->head_last_loop[es ^ 1]) PSA_HIP(hipStreamWaitEvent(st_tail, b->head_last_loop[es ^ 1], 0));
    // runs of one batch write one set of records: this run's must land after those of a run
    // without enumeration enqueued just before it; and a run without the resolve stage, whose
    // record kernel also resets the resolve accumulators, after the resolve chain before it
    if (b->last_run_res_only || (!resolve && b->res_recorded)) PSA_HIP(hipStreamWaitEvent(st_tail, b->ev_res, 0));
    b->last_run_res_only = false;
    for (int cls = psa::kEnumClasses - 1; cls >= 0; --cls) {
      const int c0 = b->cls_c0[cls + 1], c1 = b->cls_c0[cls];
      if (c1 <= c0) continue;
      P.chunks = b->d_chunks + c0;
      P.chunk_best = chunk_best + c0;
      P.nchunks = c1 - c0;
      P.fg = nullptr;
      // few workgroups, each walking many chunks: nearly every chunk is skipped after two loads,
      // and a workgroup that reserves its LDS beside the candidate loops is slow to place
      int grid = std::min({c1 - c0, std::max(64, (c1 - c0) / 16), 4 * c->num_cus});
      if (const char *e = std::getenv("PSA_LOCATE_GRID")) grid = std::max(1, std::min(c1 - c0, std::atoi(e)));  // tuning aid
      PSA_HIP(psa::launch_enum_locate(cls, P, grid, st_tail));
    }
    if (P.short_rows) {
      psa::ResolveParams R{};
      R.seq1 = c->d_seq1;
      R.len1 = c->len1;
      R.seq2_pool = b->d_pool;
      R.desc = b->d_desc;
      R.table_t = c->d_table_t;
      R.pchunks = (b->max_len2 + 255) / 256;
      R.short_rows = P.short_rows;
      R.short_total = P.short_total;
      R.short_rows_cap = P.short_rows_cap;
      R.short_quota = P.short_quota;
      R.eacc = eacc;
      // (a small batch: the record kernel picks the offsets itself, one launch less behind the candidate loop)
      const bool pick_in_records = b->nvalid <= 8;
      PSA_HIP(psa::launch_short_replay(R, b->max_len2, std::max(256, std::min(4096, b->nvalid * R.pchunks * 8)),
                                       !pick_in_records, st_tail));
      F.pick_rows = pick_in_records ? P.short_rows : nullptr;
      F.pick_rows_cap = P.short_rows_cap;
      F.pick_done = b->d_counters + 3 + par;
    }
    F.which = resolve ? PSA_STAGE_ENUMERATE : (PSA_STAGE_ENUMERATE | PSA_STAGE_RESOLVE);
    // Record kernels of runs without the resolve stage clear the float32 half of the records.  Two
    // such runs in a row end on different streams; the next resolve chain waits for the later one
    // only (enum_only_tail), so the earlier one's record kernel must not be able to land behind it:
    // the tails of consecutive runs without the resolve stage form a chain.
    if (!resolve && b->enum_only_tail) PSA_HIP(hipStreamWaitEvent(st_tail, b->enum_only_tail, 0));
    b->ev_last = b->ev_tail[par];
    PSA_HIP(psa::launch_finalize(F, b->n, st_tail, nullptr, b->ev_last));
    PSA_HIP(psa::debug_sync("psa_finalize"));
    b->tail_recorded[par] = true;
    b->tail_stream[par] = es;
    *timed = true;
    if (!resolve) b->enum_only_tail = b->ev_last;
    return PSA_OK;
  };
  if (enumerate && tail && !fast_start) {
    const int st = enqueue_tail();
    if (st != PSA_OK) return st;
  }
  if (head && (resolve || !enumerate)) {
    // the record kernel of a run without the resolve stage (the run before this one) clears the
    // float32 half of the records and resets the resolve accumulators: this chain starts after it
    if (b->enum_only_tail) PSA_HIP(hipStreamWaitEvent(c->st_res, b->enum_only_tail, 0));
    b->enum_only_tail = nullptr;
  }
  if (head && resolve) {
    psa::ResolveParams P{};
    P.seq1 = c->d_seq1;
    P.len1 = c->len1;
    P.seq2_pool = b->d_pool;
    P.desc = b->d_desc;
    P.fwd_blocks = b->d_fwd;
    P.acc = b->d_acc;
    P.table_t = c->d_table_t;
    P.fs = b->d_fs;
    P.active = b->d_active;
    P.nactive = b->d_counters;
    P.pchunks = (b->max_len2 + 255) / 256;
    P.one_seq = b->n == 1 ? 1 : 0;
    int tie_grid = std::max(512, std::min(8192, b->nvalid * P.pchunks));
    if (const char *e = std::getenv("PSA_TIE_GRID")) tie_grid = std::max(1, std::min(8192, std::atoi(e)));  // tuning aid
    PSA_HIP(psa::launch_resolve(P, b->nfwd, b->max_len2, tie_grid, c->st_res, b->ev_res_first));
  }
  if (head && (resolve || !enumerate)) {
    // the float32 half of the records is filled on the resolve stream, beside the enumeration;
    // a run without enumeration fills (clears) the other half here too
    F.which = enumerate ? PSA_STAGE_RESOLVE : (PSA_STAGE_ENUMERATE | PSA_STAGE_RESOLVE);
    if (!enumerate) {
      // ... and after those of enumerating runs still in flight, whose accumulators it also resets
      for (int i = 0; i < 2; ++i)
        if (b->tail_recorded[i]) PSA_HIP(hipStreamWaitEvent(c->st_res, b->ev_tail[i], 0));
      b->last_run_res_only = true;
    }
    PSA_HIP(psa::launch_finalize(F, b->n, c->st_res, resolve ? nullptr : b->ev_res_first, b->ev_res));
    b->res_recorded = true;
    PSA_HIP(psa::debug_sync("psa_finalize"));
  }
  if (enumerate && tail && fast_start) {
    const int st = enqueue_tail();
    if (st != PSA_OK) return st;
  }
  if (tail) b->acc_ready = true;  // psa_finalize leaves the accumulators reset
  return PSA_OK;
}

// what psa_batch_wait needs to know about the run just enqueued
static void mark_pending(psa_batch *b, uint32_t stages, bool timed, bool global_ref) {
  b->pending = true;
  b->pending_timed = timed;
  // which record kernels this run has: the enumeration side's, the resolve side's, or both
  b->pending_resolve = ((stages & PSA_STAGE_RESOLVE) && b->nfwd > 0) || !timed;
  b->pending_both = timed && (stages & PSA_STAGE_RESOLVE) && b->nfwd > 0;
  b->pending_short = timed && b->d_short != nullptr;
  b->global_ref = global_ref;
}

int psa_batch_enqueue(psa_batch *b, uint32_t stages) {
  if (!b) return PSA_ERR_ARG;
  if (!b->ctx) return PSA_ERR_STATE;  // its context was shut down
  if (stages == 0) stages = PSA_STAGE_ENUMERATE | PSA_STAGE_RESOLVE;
  psa_ctx *c = b->ctx;
  b->ran = false;
  b->pending = false;
  if (b->n == 0) {
    b->pending = true;
    return PSA_OK;
  }
  PSA_HIP(hipSetDevice(c->device));
  // (Capturing a run into a HIP graph and replaying it was measured: 0.460 against 0.463 ms per
  // run of 4 x Seq2 = 2000, and events inside a graph do not time its kernels -- not used.)
  bool timed = false;
  {
    int st = enqueue_run(b, stages, &timed);
    if (st != PSA_OK) return st;
  }
  mark_pending(b, stages, timed, false);
  return PSA_OK;
}

int psa_batch_wait(psa_batch *b) {
  if (!b) return PSA_ERR_ARG;
  if (!b->ctx || !b->pending) return PSA_ERR_STATE;
  b->pending = false;
  if (b->n == 0) {
    b->ran = true;
    return PSA_OK;
  }
  psa_ctx *c = b->ctx;
  PSA_HIP(hipSetDevice(c->device));
  // the record kernels are the last of their streams
  if (b->pending_timed) PSA_HIP(hipEventSynchronize(b->ev_last));
  if (b->pending_resolve) PSA_HIP(hipEventSynchronize(b->ev_res));
  // The run enqueued before the last one ends on the other candidate-loop stream (locate and its
  // record kernel) and nothing orders it before the last run's tail: wait for it too, so that
  // after this call no kernel of the batch is in flight (psa_batch_destroy then hands the
  // buffers to the next batch without synchronising).
  for (int i = 0; i < 2; ++i)
    if (b->tail_recorded[i]) PSA_HIP(hipEventSynchronize(b->ev_tail[i]));
  b->upload_in_flight = false;
  b->enum_ms = 0.f;
  b->total_ms = 0.f;
  if (b->pending_timed) {
    // time during which at least one candidate loop of the run was executing: consecutive
    // passes run on two streams and overlap, so their durations do not simply add up
    std::vector<std::pair<float, float>> iv;
    for (psa_batch::SlabPass &sp : b->passes) {
      float t0 = 0.f, t1 = 0.f;
      PSA_HIP(hipEventElapsedTime(&t0, b->ev_first, sp.ev_b[sp.last_buf]));
      PSA_HIP(hipEventElapsedTime(&t1, b->ev_first, sp.ev_e[sp.last_buf]));
      iv.emplace_back(t0, t1);
    }
    std::sort(iv.begin(), iv.end());
    float hi = -1e30f;
    for (const auto &x : iv) {
      if (x.first > hi) b->enum_ms += x.second - x.first;
      else if (x.second > hi) b->enum_ms += x.second - hi;
      hi = std::max(hi, x.second);
    }
  }
  // first kernel start to last kernel end over both sides
  for (hipEvent_t first : {b->pending_timed ? b->ev_first : nullptr, b->pending_resolve ? b->ev_res_first : nullptr})
    for (hipEvent_t last : {b->pending_timed ? b->ev_last : nullptr, b->pending_resolve ? b->ev_res : nullptr}) {
      float ms = 0.f;
      if (first && last && hipEventElapsedTime(&ms, first, last) == hipSuccess) b->total_ms = std::max(b->total_ms, ms);
    }
  (void)hipGetLastError();
  // Cross-check of the two stages (psa.h, enum_check): the float32 maximum and the fixed-point
  // maximum of every Seq2 are the same quantity up to the rigorous rounding bound b.
  for (int i = 0; i < b->n; ++i) {
    psa_result &r = b->h_out[i];
    r.enum_check = 0;
    if (!b->pending_both || r.status != PSA_OK || r.enum_frac_bits < 0 || r.n <= 0 || r.enum_n <= 0) continue;
    const double m = (double)b->desc[(size_t)i].len2 - 2.0, u = std::ldexp(1.0, -24);
    const double bound = (m - 1.0) * u / (1.0 - (m - 1.0) * u) * m * c->table_abs_max + m * std::ldexp(1.0, -(r.enum_frac_bits + 1));
    const double diff = std::fabs((double)r.score - std::ldexp((double)r.enum_score, -r.enum_frac_bits));
    r.enum_check = diff <= bound ? 1 : -1;
  }
  // The replayed shortlist (psa.h): in a run with both stages its winner must BE the float32 answer; a run of
  // the enumeration stage alone takes its answer from it.
  for (int i = 0; i < b->n && b->pending_short; ++i) {
    psa_result &r = b->h_out[i];
    const psa::ShortOut &so = b->h_short[i];
    if (r.status != PSA_OK || so.state != 1) continue;
    if (b->pending_both) {
      // (a shard that shortlisted against the maximum over all shards need not hold its own float32 answer
      // inside that window: the replayed winners are compared after the combine, psa_batch_run_allgather)
      if (r.enum_check == 1 && !b->global_ref)
        r.enum_check = (so.score_bits == r.score_bits && so.n == r.n && so.k == r.k && so.offset == r.offset) ? 2 : -2;
    } else if (so.n > 0) {  // (a shard's list against the maximum over all shards can be complete and empty)
      r.n = so.n;
      r.k = so.k;
      r.offset = so.offset;
      r.score_bits = so.score_bits;
      std::memcpy(&r.score, &so.score_bits, sizeof(float));
      r.mutant_num = (int32_t)(psa::mutant_of_nk(so.n, so.k, b->desc[(size_t)i].len2) - 1);
      for (int k = 0; k < 4; ++k) r.counts[k] = so.counts[k];
    }
  }
  b->ran = true;
  return PSA_OK;
}

int psa_batch_run(psa_batch *b, uint32_t stages) {
  int st = psa_batch_enqueue(b, stages);
  return st == PSA_OK ? psa_batch_wait(b) : st;
}

int psa_batch_results(const psa_batch *b, psa_result *out) {
  if (!b || (!out && b->n > 0)) return PSA_ERR_ARG;
  if (!b->ctx || !b->ran) return PSA_ERR_STATE;
  if (b->n > 0) std::memcpy(out, b->h_out, (size_t)b->n * sizeof(psa_result));
  return PSA_OK;
}

int psa_batch_shortlist(const psa_batch *b, int32_t *count, int32_t *state) {
  if (!b) return PSA_ERR_ARG;
  if (!b->ctx || !b->ran) return PSA_ERR_STATE;
  for (int i = 0; i < b->n; ++i) {
    const bool have = b->pending_short && b->h_out[i].status == PSA_OK;
    if (count) count[i] = have ? b->h_short[i].count : 0;
    if (state) state[i] = have ? b->h_short[i].state : 0;
  }
  return PSA_OK;
}

int psa_batch_timing(const psa_batch *b, float *enum_kernel_ms, float *total_ms) {
  if (!b || !b->ran) return PSA_ERR_STATE;
  if (enum_kernel_ms) *enum_kernel_ms = b->enum_ms;
  if (total_ms) *total_ms = b->total_ms;
  return PSA_OK;
}

int psa_batch_event_times(const psa_batch *b, const psa_batch *ref, float ms[6]) {
  if (!b || !ref || !ms) return PSA_ERR_ARG;
  if (!b->ran || !ref->ran || b->ctx != ref->ctx || !b->ctx) return PSA_ERR_STATE;
  hipEvent_t origin = ref->pending_timed ? ref->ev_first : ref->ev_res_first;
  for (int i = 0; i < 6; ++i) ms[i] = -1.f;
  if (!origin) return PSA_ERR_STATE;
  auto at = [&](hipEvent_t e, float *out) {
    float t = 0.f;
    if (e && hipEventElapsedTime(&t, origin, e) == hipSuccess) *out = t;
  };
  if (b->pending_timed && !b->passes.empty()) {
    const psa_batch::SlabPass &first = b->passes.front(), &last = b->passes.back();
    at(b->ev_first, &ms[0]);
    at(first.ev_b[first.last_buf], &ms[1]);
    at(last.ev_e[last.last_buf], &ms[2]);
    at(b->ev_last, &ms[3]);
  }
  if (b->pending_resolve) {
    at(b->ev_res_first, &ms[4]);
    at(b->ev_res, &ms[5]);
  }
  (void)hipGetLastError();
  return PSA_OK;
}

int psa_score_batch(psa_ctx *c, int n, const char *const *seq2, const int32_t *len2, uint32_t stages,
                    psa_result *out) {
  psa_batch *b = nullptr;
  if (stages == 0) stages = PSA_STAGE_ENUMERATE | PSA_STAGE_RESOLVE;
  int st = batch_build(c, n, seq2, len2, false, 0, 0, stages, &b);
  if (st != PSA_OK) return st;
  st = psa_batch_run(b, stages);
  if (st == PSA_OK) st = psa_batch_results(b, out);
  psa_batch_destroy(b);
  return st;
}

int psa_score_range(psa_ctx *c, const char *seq2, int32_t len2, int64_t s_begin, int64_t s_end, uint32_t stages,
                    psa_result *out) {
  psa_batch *b = nullptr;
  if (stages == 0) stages = PSA_STAGE_ENUMERATE | PSA_STAGE_RESOLVE;
  const char *one[1] = {seq2};
  int32_t l[1] = {len2};
  int st = batch_build(c, 1, one, len2 >= 0 ? l : nullptr, true, s_begin, s_end, stages, &b);
  if (st != PSA_OK) return st;
  st = psa_batch_run(b, stages);
  if (st == PSA_OK) st = psa_batch_results(b, out);
  psa_batch_destroy(b);
  return st;
}

// ---- the answer path chosen by the host program ---------------------------------------
int psa_parse_stages(const char *text, uint32_t *stages) {
  if (!stages) return PSA_ERR_ARG;
  if (!text || !*text || std::strcmp(text, "resolve") == 0) *stages = PSA_STAGE_RESOLVE;
  else if (std::strcmp(text, "enumerate") == 0) *stages = PSA_STAGE_ENUMERATE;
  else if (std::strcmp(text, "both") == 0) *stages = PSA_STAGE_ENUMERATE | PSA_STAGE_RESOLVE;
  else {
    g_err = std::string("stages must be resolve, enumerate or both, not '") + text + "'";
    return PSA_ERR_ARG;
  }
  return PSA_OK;
}

int psa_score_batch_answer(psa_ctx *c, int n, const char *const *seq2, const int32_t *len2, uint32_t stages,
                           psa_result *out, int *fallbacks) {
  if (fallbacks) *fallbacks = 0;
  if (stages == 0) stages = PSA_STAGE_ENUMERATE | PSA_STAGE_RESOLVE;
  if (stages != PSA_STAGE_ENUMERATE) {
    int st = psa_score_batch(c, n, seq2, len2, stages, out);
    if (st != PSA_OK || !(stages & PSA_STAGE_ENUMERATE)) return st;
    for (int i = 0; i < n; ++i)
      if (out[i].status == PSA_OK && out[i].enum_check < 0) {
        g_err = "Seq2 #" + std::to_string(i + 1) + ": the enumeration stage and the resolve stage disagree (enum_check " +
                std::to_string(out[i].enum_check) + ")";
        return PSA_ERR_CHECK;
      }
    return PSA_OK;
  }
  // the enumeration stage alone: the answer comes from the replayed shortlist ...
  psa_batch *b = nullptr;
  int st = batch_build(c, n, seq2, len2, false, 0, 0, PSA_STAGE_ENUMERATE, &b);
  if (st != PSA_OK) return st;
  std::vector<int32_t> state((size_t)std::max(n, 1), 0);
  st = psa_batch_run(b, PSA_STAGE_ENUMERATE);
  if (st == PSA_OK) st = psa_batch_results(b, out);
  if (st == PSA_OK) st = psa_batch_shortlist(b, nullptr, state.data());
  psa_batch_destroy(b);
  if (st != PSA_OK) return st;
  // ... and where there is none (the list overflowed; the table does not fit the fixed-point range; the
  // shortlist is switched off) from the resolve stage, for those Seq2 only
  std::vector<int> idx;
  for (int i = 0; i < n; ++i)
    if (out[i].status == PSA_OK && out[i].candidates > 0 && state[(size_t)i] != 1) idx.push_back(i);
  if (idx.empty()) return PSA_OK;
  std::vector<const char *> s2(idx.size());
  std::vector<int32_t> l2(idx.size());
  for (size_t k = 0; k < idx.size(); ++k) {
    s2[k] = seq2[idx[k]];
    l2[k] = len2 ? len2[idx[k]] : (int32_t)std::strlen(seq2[idx[k]]);
  }
  std::vector<psa_result> r(idx.size());
  st = psa_score_batch(c, (int)idx.size(), s2.data(), l2.data(), PSA_STAGE_RESOLVE, r.data());
  if (st != PSA_OK) return st;
  for (size_t k = 0; k < idx.size(); ++k) {
    psa_result &o = out[idx[k]];
    const psa_result &f = r[k];
    if (f.status != PSA_OK) {
      o.status = f.status;
      continue;
    }
    o.n = f.n;
    o.k = f.k;
    o.offset = f.offset;
    o.mutant_num = f.mutant_num;
    o.score = f.score;
    o.score_bits = f.score_bits;
    std::memcpy(o.counts, f.counts, sizeof o.counts);
  }
  if (fallbacks) *fallbacks = (int)idx.size();
  return PSA_OK;
}

// ---- one long pair over several ranks ----------------------------------------------
// The reference has no such mode (it calls calc_best_score_CUDA once per Seq2, P/main.c:81,
// :130, and each call walks all offsets); here the offsets [0, S) of one pair are cut into
// nranks contiguous shards, this rank scores its shard on its GPU, the caller's all-gather
// moves the shards' answers, and every rank combines them to the same record.
namespace {

// first exchange of an enumerating run: the shard's enumerated maximum (fixed point; every shard has the
// same Seq2 and table, hence the same scale)
struct ShardMax {
  int32_t have;   // this shard enumerated candidates
  int32_t score;
  int32_t status;
  int32_t pad_;
};
// last exchange: the shard's record and the winner of its replayed shortlist
struct ShardMsg {
  psa_result rec;
  psa::ShortOut so;
};
static_assert(sizeof(ShardMax) == PSA_EXCHANGE_MAX_BYTES, "psa.h states this size");
static_assert(sizeof(ShardMsg) == PSA_EXCHANGE_RECORD_BYTES, "psa.h states this size");

bool replay_better(const psa::ShortOut &a, const psa::ShortOut &b) {  // highest score, then smallest (n, k, offset)
  float fa, fb;
  std::memcpy(&fa, &a.score_bits, sizeof fa);
  std::memcpy(&fb, &b.score_bits, sizeof fb);
  if (fa != fb) return fa > fb;
  if (a.n != b.n) return a.n < b.n;
  if (a.k != b.k) return a.k < b.k;
  return a.offset < b.offset;
}

// The run of one shard and the exchanges around it.  b may be null and st0 an error: the rank then only
// takes part in the collectives (the other ranks are already waiting in them) and reports its status.
int run_shard_allgather(psa_batch *b, int st0, int nranks, psa_allgather_fn allgather, void *comm, uint32_t stages,
                        psa_result *out) {
  if (stages == 0) stages = PSA_STAGE_ENUMERATE | PSA_STAGE_RESOLVE;
  int st = st0;
  if (st == PSA_OK && !b->ctx) st = PSA_ERR_STATE;
  if (st == PSA_OK && b->n != 1) {
    g_err = "psa_batch_run_allgather: the batch must hold exactly one Seq2 (one shard of one pair)";
    st = PSA_ERR_ARG;
  }
  // (one rank with a callback runs the same protocol over a one-rank all-gather: a rehearsal of the transport)
  const bool exchange = nranks > 1 || allgather != nullptr;
  const bool two_parts = exchange && (stages & PSA_STAGE_ENUMERATE);
  if (!two_parts) {
    if (st == PSA_OK) st = psa_batch_run(b, stages);
  } else {
    // candidate loops (and the resolve chain beside them); then the shards' enumerated maxima go round, and
    // locate shortlists this shard against the maximum over all of them
    ShardMax mx{};
    bool timed = false;
    if (st == PSA_OK) {
      psa_ctx *c = b->ctx;
      b->ran = b->pending = false;
      auto head = [&]() -> int {
        PSA_HIP(hipSetDevice(c->device));
        int e = enqueue_run(b, stages, &timed, kPartHead);
        if (e != PSA_OK) return e;
        if ((stages & PSA_STAGE_ENUMERATE) && b->nchunks > 0) {
          for (hipEvent_t ev : b->head_last_loop)
            if (ev) PSA_HIP(hipEventSynchronize(ev));
          unsigned long long key = 0ull;
          const psa::SeqAcc *eacc = b->d_acc + (size_t)b->acc_par * (size_t)b->n;
          PSA_HIP(hipMemcpy(&key, &eacc->enum_key, sizeof key, hipMemcpyDeviceToHost));
          if (key != 0ull) {
            mx.have = 1;
            mx.score = (int32_t)((uint32_t)(key >> 32) ^ 0x80000000u);
          }
        }
        return PSA_OK;
      };
      st = head();
      if (st != PSA_OK) b->acc_ready = false;
    }
    mx.status = st;
    std::vector<ShardMax> all((size_t)nranks);
    const bool comm_ok = allgather(&mx, all.data(), sizeof mx, comm) == 0;
    int gref = INT32_MIN;
    bool have = false;
    for (const ShardMax &a : all)
      if (comm_ok && a.have) {
        gref = have ? std::max(gref, a.score) : a.score;
        have = true;
      }
    if (st == PSA_OK) {
      // (the second part is enqueued even after a failed exchange: it leaves the batch's accumulators reset)
      st = enqueue_run(b, stages, &timed, kPartTail, have ? &gref : nullptr);
      if (st == PSA_OK) {
        mark_pending(b, stages, timed, have);
        st = psa_batch_wait(b);
      } else {
        b->acc_ready = false;
      }
    }
    if (!comm_ok) {
      g_err = "psa_batch_run_allgather: the caller's all-gather reported a failure";
      return PSA_ERR_COMM;
    }
  }
  ShardMsg mine{};
  if (st == PSA_OK) st = psa_batch_results(b, &mine.rec);
  if (st == PSA_OK && b->pending_short) mine.so = b->h_short[0];
  if (st != PSA_OK) {
    mine = ShardMsg{};
    mine.rec.status = st;
  }
  if (!exchange) {  // what the combine returns for one part
    *out = mine.rec;
    return st != PSA_OK ? st : mine.rec.status;
  }
  std::vector<ShardMsg> msgs((size_t)nranks);
  if (allgather(&mine, msgs.data(), sizeof(ShardMsg), comm) != 0) {
    g_err = "psa_batch_run_allgather: the caller's all-gather reported a failure";
    return PSA_ERR_COMM;
  }
  std::vector<psa_result> parts((size_t)nranks);
  for (int i = 0; i < nranks; ++i) parts[(size_t)i] = msgs[(size_t)i].rec;
  st = psa_combine_results(parts.data(), nranks, out);
  if (st != PSA_OK || !(stages & PSA_STAGE_ENUMERATE)) return st;
  // The replayed shortlists of all shards together hold the reference's answer when every one of them is
  // complete (each hangs below the maximum over all shards): their best winner against the combined record.
  bool complete = true;
  int best = -1;
  for (int i = 0; i < nranks; ++i) {
    const ShardMsg &m = msgs[(size_t)i];
    if (m.rec.candidates == 0) continue;  // a shard without an offset
    if (m.rec.enum_frac_bits < 0 || m.so.state != 1) {
      complete = false;
      continue;
    }
    if (m.so.n > 0 && (best < 0 || replay_better(m.so, msgs[(size_t)best].so))) best = i;
  }
  if (stages & PSA_STAGE_RESOLVE) {
    if (out->enum_check == 1 && complete && out->n > 0) {
      const psa::ShortOut *w = best >= 0 ? &msgs[(size_t)best].so : nullptr;
      out->enum_check = (w && w->score_bits == out->score_bits && w->n == out->n && w->k == out->k && w->offset == out->offset) ? 2 : -2;
    }
  } else if (!complete || best < 0) {
    // enumeration alone and some shard's list overflowed: no float32 answer (run the resolve stage)
    out->n = out->k = out->offset = out->mutant_num = 0;
    out->score = 0.0f;
    out->score_bits = 0u;
    std::memset(out->counts, 0, sizeof out->counts);
  }
  return PSA_OK;
}

}  // namespace

int psa_batch_run_allgather(psa_batch *b, int nranks, psa_allgather_fn allgather, void *comm, uint32_t stages,
                            psa_result *out) {
  if (!b || !out || nranks < 1 || (nranks > 1 && !allgather)) return PSA_ERR_ARG;
  return run_shard_allgather(b, PSA_OK, nranks, allgather, comm, stages, out);
}

int psa_score_range_allgather(psa_ctx *c, const char *seq2, int32_t len2, int rank, int nranks,
                              psa_allgather_fn allgather, void *comm, uint32_t stages, psa_result *out) {
  // (errors of the call itself, the same on every rank: nobody enters a collective)
  if (!c || !seq2 || !out || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !allgather))
    return PSA_ERR_ARG;
  if (stages == 0) stages = PSA_STAGE_ENUMERATE | PSA_STAGE_RESOLVE;
  psa_batch *b = nullptr;
  int st = PSA_OK;
  // A context without Seq1 or table is this rank's own trouble: it still enters the collectives below, so
  // that the other ranks do not wait for it forever, and all ranks return its status.
  if (!c->have_seq1 || !c->have_table) {
    g_err = "psa_score_range_allgather: Seq1 or the table is not set";
    st = PSA_ERR_STATE;
  } else {
    const int64_t L2 = len2 >= 0 ? (int64_t)len2 : (int64_t)std::strlen(seq2);
    const int64_t S = c->len1 - L2 + 3;  // P/cudaFunctions.cu:269
    // (an invalid pair is invalid for every rank alike: any range lets the batch record the status)
    const int64_t lo = S >= 1 ? S * rank / nranks : 0, hi = S >= 1 ? S * (rank + 1) / nranks : 1;
    const char *one[1] = {seq2};
    int32_t l[1] = {len2};
    // (laid out for the stages that run: a resolve-only pair builds no chunk list, keys or slabs)
    st = batch_build(c, 1, one, len2 >= 0 ? l : nullptr, true, lo, hi, stages, &b);
  }
  st = run_shard_allgather(b, st, nranks, allgather, comm, stages, out);
  if (b) psa_batch_destroy(b);
  return st;
}

// ---- which GPU a process started by a launcher should use ----------------------------
// PSA_DEVICE if set; otherwise this process's rank on its node as the launcher exports it
// (Open MPI, MVAPICH2, MPICH/Intel MPI hydra, PMI, Slurm, torch.distributed.run), modulo the
// number of visible GPUs; otherwise 0.  The reference never calls cudaSetDevice: each of its
// two ranks uses device 0 of its own host (P/cudaFunctions.cu:260-323, P/myHosts).
int psa_default_device(void) {
  const int ngpu = psa_device_count();
  if (const char *e = std::getenv("PSA_DEVICE")) return std::atoi(e);
  static const char *const kLocalRankVars[] = {"OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK",
                                               "MPI_LOCALRANKID", "PMI_LOCAL_RANK", "SLURM_LOCALID", "LOCAL_RANK"};
  for (const char *name : kLocalRankVars)
    if (const char *e = std::getenv(name)) {
      char *end = nullptr;
      const long v = std::strtol(e, &end, 10);
      if (end != e && v >= 0) return ngpu > 0 ? (int)(v % ngpu) : 0;
    }
  return 0;
}

// ---- drop-in for calc_best_score_CUDA (P/cudaFunctions.cu:260-323) ----------------
// Keeps one context alive; Seq1 and the table are uploaded again only when they
// change between calls (the reference re-uploads both for every Seq2 and never frees).
char *psa_calc_best_score(const char *seq1, const char *seq2, const float *comp_matrix) {
  // One context per process, shared by every caller: calls are serialised (the reference calls
  // from one thread per rank, P/main.c:79-84).
  static std::mutex mu;
  static psa_ctx *ctx = nullptr;
  static std::string cached_seq1;
  static float cached_table[PSA_TABLE_SIZE];
  static bool have_table = false;
  auto die = [](const char *what, int st) {
    std::fprintf(stderr, "%s: %s (%s)\n", what, psa_strerror(st), psa_last_error_text());
    std::exit(EXIT_FAILURE);
  };
  if (!seq1 || !seq2 || !comp_matrix) die("psa_calc_best_score", PSA_ERR_ARG);
  std::lock_guard<std::mutex> lock(mu);
  int st;
  if (!ctx) {
    const int dev = psa_default_device();
    if (std::getenv("PSA_DEBUG_SYNC"))
      std::fprintf(stderr, "[psa] psa_calc_best_score: pid %d uses GPU %d of %d\n", (int)getpid(), dev,
                   psa_device_count());
    if ((st = psa_init(dev, &ctx)) != PSA_OK) die("psa_init", st);
  }
  if (cached_seq1 != seq1) {
    if ((st = psa_set_seq1(ctx, seq1, -1)) != PSA_OK) die("psa_set_seq1", st);
    cached_seq1 = seq1;
  }
  if (!have_table || std::memcmp(cached_table, comp_matrix, sizeof cached_table) != 0) {
    if ((st = psa_set_table(ctx, comp_matrix)) != PSA_OK) die("psa_set_table", st);
    std::memcpy(cached_table, comp_matrix, sizeof cached_table);
    have_table = true;
  }
  psa_result r{};
  const char *one[1] = {seq2};
  // PSA_STAGES = resolve (default: the float32 recurrence gives the reference's answer by itself), enumerate
  // (every candidate scored by the candidate loop, as every call of the reference does,
  // P/cudaFunctions.cu:290-296; the answer from the replayed shortlist, the resolve stage only where that
  // overflowed) or both (the two must agree, or the call dies)
  uint32_t stages = PSA_STAGE_RESOLVE;
  if ((st = psa_parse_stages(std::getenv("PSA_STAGES"), &stages)) != PSA_OK) die("PSA_STAGES", st);
  if ((st = psa_score_batch_answer(ctx, 1, one, nullptr, stages, &r, nullptr)) != PSA_OK) die("psa_score_batch_answer", st);
  if (r.status != PSA_OK) die("psa_calc_best_score", r.status);
  char *line = psa_format_result(&r);
  if (!line) die("psa_format_result", PSA_ERR_NOMEM);
  const char *quiet = std::getenv("PSA_SHIM_QUIET");
  if (!(quiet && quiet[0] == '1')) std::printf("%s", line);  // P/cudaFunctions.cu:321
  return line;
}

}  // extern "C"
