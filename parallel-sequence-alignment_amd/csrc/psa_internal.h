// Internal declarations shared by the host helpers, the kernels and the C ABI.
#ifndef PSA_INTERNAL_H
#define PSA_INTERNAL_H

#include <stdint.h>

#include "../../include/psa.h"

namespace psa {

// Device tables are stored with a row stride of 32 and 27 rows: code 26 is the
// padding residue (positions outside a sequence); its row and column are zero.
// They are indexed [seq2 code][seq1 code] so that lanes which differ only in the
// Seq1 residue read one 27-word row.
constexpr int kRowStride = 32;
constexpr int kPadCode = 26;
constexpr int kDevTable = 27 * kRowStride;  // 864 entries

// One Seq2 of a batch as the kernels see it.
struct SeqDesc {
  int64_t fs_base;    // index of this Seq2's first entry in the per-offset score array
  int32_t off2;       // byte offset of its residue codes in the Seq2 pool
  int32_t len2;
  int32_t s_begin;    // offsets [s_begin, s_end) are scored
  int32_t s_end;
  int32_t frac_bits;  // fixed-point scale of the enumeration stage, -1 = stage off for this Seq2
  int32_t enum_ns;    // offsets one wave of the enumeration kernel walks (a sub-run of a chunk)
  int32_t enum_slack; // shortlist window below the enumerated maximum in fixed-point units, -1 = no shortlist
  int32_t pad_;
};

// Per-Seq2 accumulators the kernels reduce into with 64/32-bit atomics.
struct SeqAcc {
  unsigned long long enum_key;  // max of (biased fixed-point score << 32 | ~p)
  unsigned long long enum_loc;  // min of (q << 32 | offset) among candidates equal to the maximum with p = p*
  unsigned long long res_loc;   // min of (q << 32 | offset) for the float32 path
  uint32_t res_key;             // max of the order-preserving image of the float32 score
  int32_t res_p;                // min p attaining the float32 maximum
  unsigned long long short_best;  // shortlist: max of (float32 score image << 32 | ~p << 16 | ~q) over the replayed rows
  uint32_t short_s;             // ... and the smallest offset among the rows that attain it
  uint32_t short_count;         // rows appended for this Seq2 (beyond the batch's quota: given up)
};

// Shortlist of the enumeration stage.  The candidate the reference reports has a fixed-point score
// within SeqDesc::enum_slack of the enumerated maximum (twice the rigorous bound between the
// fixed-point and the float32 left-to-right sum), so every (offset, p) that has such a candidate is
// listed as a row, with the 256-wide blocks of q that hold one, and the listed blocks are replayed
// in float32 exactly as P/cudaFunctions.cu:210-220 adds them.  One list per run for the whole
// batch, with room for every Seq2's full quota of rows: a Seq2 with more rows than the quota is given
// up (a low-entropy input can put every row inside the window), and whether it is depends on nothing
// but its own rows -- not on the other Seq2 of the batch, nor on the order in which blocks append.
struct ShortRow {
  int32_t seq;
  int32_t s;
  uint32_t p;
  uint32_t qmask;           // bit c: some q in [256 c, 256 c + 256) lies inside the window
  unsigned long long best;  // replay: max over the row's q of (score image << 32 | ~q)
};
constexpr int kShortRowsPerSeq = 1024;     // the quota, unless the batch is so large that ...
constexpr int kShortBudgetRows = 1 << 24;  // ... its list would pass this many rows (24 bytes each, two copies): then budget / Seq2
// What the record kernel reports of a Seq2's shortlist (pinned host memory, beside the records).
struct ShortOut {
  uint32_t score_bits;  // the replayed winner: highest float32 score, ties to the smallest (n, k, offset)
  int32_t n, k, offset;
  int32_t counts[4];
  int32_t count;        // rows replayed
  int32_t state;        // 0: no shortlist in this run, 1: complete, 2: given up (the winner fields are void)
};
constexpr int kShortPList = 256;  // most positions p of one offset that may reach the window

// ns consecutive offsets of one Seq2: the unit of the enumeration pre-pass, which writes
// the g values of those offsets (and of the offset after them: f of an offset is read from the
// next offset's row) as ns + 1 rows of the slab starting at fg_row.
struct Chunk {
  int32_t seq;
  int32_t s0;
  int32_t ns;
  uint32_t fg_row;  // first row of the chunk in the slab, in units of 64 int32
};

// A block of up to kFwdOffsets consecutive offsets of one Seq2 for the forward recurrence
// (4 waves x 62: two lanes of every wave only feed their neighbours).
constexpr int kFwdOffsets = 248;
struct FwdBlock {
  int32_t seq;
  int32_t s0;
};

struct ActiveOffset {
  int32_t seq;
  int32_t s;
};

int residue_class_code(int a, int b);
void build_table_from_weights(const float w[4], float out[PSA_TABLE_SIZE]);
constexpr int kSumBitsI32 = 29;  // prefix-sum width of the enumeration stage in int32 arithmetic
constexpr int kSumBitsF32 = 23;  // and in float32 arithmetic (integers of at most 24 bits)
int choose_frac_bits(const float table[PSA_TABLE_SIZE], int64_t len2, int sum_bits, int *exact_level);
// the same in two steps: a scan of the table per candidate scale (once per table), then the choice per Seq2
struct FracScan {
  bool ok[25];         // every entry times 2^r rounds to a representable integer
  bool exact[25];      // ... and equals the entry exactly
  int64_t mx[25];      // largest |round(entry * 2^r)|
  int min_tz[25];      // fewest trailing zero bits among the non-zero rounded entries
};
void frac_table_scan(const float table[PSA_TABLE_SIZE], FracScan *fs);
int choose_frac_bits(const FracScan &fs, int64_t len2, int sum_bits, int *exact_level);
void nk_of_mutant(int64_t t, int len2, int *n, int *k);
int64_t mutant_of_nk(int n, int k, int len2);
double cost_estimate(int64_t len2, int64_t len1, uint32_t stages);
int partition_by_cost(int n, const int32_t *len2, int64_t len1, int nparts, uint32_t stages, int32_t *owner);

}  // namespace psa

#endif
