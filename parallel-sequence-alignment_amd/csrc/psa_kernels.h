// Kernel parameter blocks and launch entry points (psa_kernels.hip <-> psa_api.hip).
#ifndef PSA_KERNELS_H
#define PSA_KERNELS_H

#include <hip/hip_runtime.h>

#include "psa_internal.h"

namespace psa {

constexpr int kEnumMaxNs = 64;      // most offsets in one enumeration chunk
constexpr int kEnumClasses = 6;     // (threads, registers) variants of the pre-pass and locate kernels
constexpr int kEnumWaveRuns = 4;    // runs of 64 positions p one wave of psa_enum_wave keeps in registers
constexpr int kEnumMaxLevels = 32;  // ceil(ceil(PSA_MAX_SEQ2 / 64) / kEnumWaveRuns)

// Pre-pass and locate kernels: one workgroup per chunk.
struct EnumParams {
  const unsigned char *seq1;       // residue codes 0..25
  int64_t len1;
  const unsigned char *seq2_pool;  // residue codes of every Seq2 of the batch
  const SeqDesc *desc;
  const Chunk *chunks;             // chunks of this launch
  int nchunks;
  SeqAcc *acc;
  const float *table_t;            // float table [seq2 code][seq1 code], stride 32, 27 rows
  unsigned long long *chunk_best;  // per chunk: max of its waves' (maximum, smallest p) keys
  int *fg;                         // slab the pre-pass writes: per offset 64*nruns f values, then 64*nruns g values
  ShortRow *short_rows;            // locate: this run's shortlist (null: none), ...
  unsigned *short_total;           // ... rows appended so far ...
  int short_rows_cap;              // ... and the list's capacity: Seq2 of the batch x short_quota, so it never fills
  int short_quota;                 // rows one Seq2 may list before it is given up
  // locate, one shard of a long pair: the shortlist window hangs below this fixed-point score (the maximum
  // over ALL shards, exchanged between the ranks after the candidate loops) instead of the shard's own
  int has_short_ref;
  int short_ref;
  // pre-pass: one wave per chunk (throughput: batches of many slab passes) instead of one block per chunk
  // (latency: a small batch's few hundred chunks, where the next candidate loop waits for this kernel)
  int prep_wave_form;
};

// psa_enum_wave: one unit of work per (chunk, sub-run of its offsets, level).  Level w owns the
// positions p in [256 w, 256 w + 256); chunks are sorted by falling Seq2 length, so the
// chunks that have a level w are a prefix of the launch's chunk list and the units are
// numbered level by level: unit u of level w is sub-run (u - level_base[w]) of that prefix.
struct WaveParams {
  const Chunk *chunks;
  const SeqDesc *desc;
  SeqAcc *acc;
  unsigned long long *chunk_best;
  int sub_log2;                            // sub-runs per chunk = 1 << sub_log2
  int level_base[kEnumMaxLevels + 1];      // first unit of each level; unused entries hold INT_MAX
  int nunits;
  // Banded numbering (band > 0): workgroups are dealt to the 8 XCDs round-robin, each XCD has its own L2, and a
  // row of g is read by every level of its chunk.  The levels are taken in bands of `band` adjacent ones, band
  // after band (largest units first, as in level-major numbering); within a band, unit r belongs to column
  // (sub-run of a chunk) 8 (r / (8 band)) + r % 8 and level r / 8 % band of the band: the band's levels of a
  // column run on ONE XCD, side by side, and read the column's rows from that XCD's L2.  Columns past ncols
  // and levels past nlev (the counts are rounded up to 8 and to the band) are empty units.
  int band;      // 0: level-major numbering
  int nlev;      // levels of the launch's longest Seq2
  int ncols;     // sub-runs of all chunks of the launch
  int ncols8;    // ncols rounded up to a multiple of 8
  // diagnostic (null in normal runs): 64 slots of 8 words; a unit adds to slot (unit & 63):
  // [0] += shader cycles, [1] += 100 MHz ticks, [2] += 1
  unsigned long long *clock_probe;
};

struct ResolveParams {
  const unsigned char *seq1;
  int64_t len1;
  const unsigned char *seq2_pool;
  const SeqDesc *desc;
  const FwdBlock *fwd_blocks;
  SeqAcc *acc;
  const float *table_t;
  float *fs;             // per-offset maxima, SeqDesc::fs_base indexes it
  ActiveOffset *active;  // offsets whose maximum equals their Seq2's maximum
  unsigned *nactive;
  int pchunks;           // ceil(max len2 / 256)
  int one_seq;           // the batch holds a single Seq2: what is known about it prunes whole blocks of the tie kernels
  // replay of the enumeration stage's shortlist (launch_short_replay only)
  ShortRow *short_rows;
  const unsigned *short_total;
  int short_rows_cap;
  int short_quota;
  SeqAcc *eacc;          // the run's copy of the enumeration accumulators
};

struct FinalParams {
  const unsigned char *seq1;
  const unsigned char *seq2_pool;
  const SeqDesc *desc;
  SeqAcc *acc;                   // float32 (resolve) accumulators: read, then reset for the next run
  SeqAcc *eacc;                  // enumeration accumulators of this run (one of two copies): likewise
  unsigned *nactive;             // reset with the resolve half
  const unsigned char *class_t;  // residue class [seq2 code][seq1 code], stride 32
  psa_result *out;               // pinned host memory, written directly
  uint32_t stages;  // stages of the run
  uint32_t which;   // PSA_STAGE_* bits: the halves of the records this launch fills
  unsigned *short_total;  // this run's shortlist row counter (null: no shortlist): reset by the enumeration half ...
  ShortOut *short_out;    // ... which reports the replayed winners here (pinned host memory)
  int short_quota;        // rows one Seq2 may list before it is given up
  int short_empty_ok;     // a list without a row is complete (state 1, no winner): shards of a long pair, see EnumParams::short_ref
  // small batches: the enumeration half also picks each Seq2's smallest offset among its winning rows
  // (psa_short_pick's job: one launch less behind the candidate loop); null otherwise
  const ShortRow *pick_rows;
  int pick_rows_cap;
  unsigned *pick_done;    // blocks that have read the list: the last one resets its row counter
};

int enum_class_of(int len2);          // -1 if len2 is too long
int enum_class_positions(int c);      // positions (= threads x registers) the class covers
int enum_class_lds_bytes(int c);
int enum_class_threads(int c);

hipError_t debug_sync(const char *what);
// dynamic-LDS limits of every kernel variant on the current device (per device, so per context)
hipError_t set_enum_kernel_attributes();
hipError_t set_resolve_kernel_attributes();
hipError_t launch_init(SeqAcc *acc, int n, unsigned *counters, int ncounters, hipStream_t st);
// set-up of a small batch in one launch: copy of the staged work lists (src: device mapping of pinned
// host memory; bytes rounded up to 16 must lie inside both blocks) + what launch_init does + cleared chunk keys
hipError_t launch_setup_small(const void *src, void *dst, size_t bytes, SeqAcc *acc, int nacc, unsigned *counters,
                              int ncounters, unsigned long long *best, size_t nbest, hipStream_t st);
// ev_start / ev_stop (may be null) are attached to the kernel's dispatch, not recorded beside it
// f32: the slab holds float32 values and the candidate loop adds and compares in float32
// (exact: the scale keeps every value an integer of at most 24 bits); otherwise int32
hipError_t launch_enum_prep(int cls, const EnumParams &P, bool f32, hipStream_t st, hipEvent_t ev_start,
                            hipEvent_t ev_stop);
// (workgroups of waves_per_group waves, 1 to 8, one unit per wave; lds_bytes are requested per workgroup and never touched)
// (use_mfma: the matrix-core form of the loop, float32 only; else the vector-ALU form)
hipError_t launch_enum_wave(const int *fg, const WaveParams &P, bool f32, bool use_mfma, int waves_per_group, int lds_bytes,
                            hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop);
hipError_t launch_enum_locate(int cls, const EnumParams &P, int grid, hipStream_t st);
// float32 replay of the shortlist rows that locate appended (P.short_rows, P.short_total, P.eacc), then the
// smallest offset among the rows that attain each Seq2's best
hipError_t launch_short_replay(const ResolveParams &P, int max_len2, int grid, bool pick, hipStream_t st);
hipError_t launch_resolve(const ResolveParams &P, int nfwd_blocks, int max_len2, int tie_grid, hipStream_t st,
                          hipEvent_t ev_start);
hipError_t launch_finalize(const FinalParams &P, int n, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop);

}  // namespace psa

#endif
