// Host-side helpers of libpsa_hip: the residue-class table, the table builder,
// mutant numbering, result formatting and the work split.  No GPU code here.
// Citations: P/ = "Parallel Solution/", S/ = "Sequential Solution/".
#include "psa_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

namespace psa {

// Group memberships as 26-bit masks (bit i = letter 'A'+i).  Two different
// letters are conservative if some conservative group holds both, otherwise
// semi-conservative if some semi-conservative group holds both
// (P/cFunctions.c:84-100); identity is tested first (P/cFunctions.c:68).
static uint32_t mask_of(const char *letters) {
  uint32_t m = 0;
  for (; *letters; ++letters) m |= 1u << (*letters - 'A');
  return m;
}

struct ClassMasks {
  uint32_t cons[26];
  uint32_t semi[26];
  ClassMasks() {
    static const char *const kCons[] = {"NDEQ", "NEQK", "STA", "MILV", "QHRK", "NHQK", "FYW", "HY", "MILF"};
    static const char *const kSemi[] = {"SAG",    "ATV",    "CSA",    "SGND", "STPA", "STNK",
                                        "NEQHRK", "NDEQHK", "SNDEQK", "HFY",  "FVLIM"};
    std::memset(cons, 0, sizeof cons);
    std::memset(semi, 0, sizeof semi);
    for (const char *g : kCons) {
      uint32_t m = mask_of(g);
      for (int a = 0; a < 26; ++a)
        if (m >> a & 1) cons[a] |= m;
    }
    for (const char *g : kSemi) {
      uint32_t m = mask_of(g);
      for (int a = 0; a < 26; ++a)
        if (m >> a & 1) semi[a] |= m;
    }
  }
};

int residue_class_code(int a, int b) {
  static const ClassMasks masks;
  if (a == b) return 0;
  if (a < 0 || a > 25 || b < 0 || b > 25) return 3;
  if (masks.cons[a] >> b & 1) return 1;
  if (masks.semi[a] >> b & 1) return 2;
  return 3;
}

void build_table_from_weights(const float w[4], float out[PSA_TABLE_SIZE]) {
  for (int a = 0; a < 26; ++a)
    for (int b = 0; b < 26; ++b) {
      int c = residue_class_code(a, b);
      out[a * 26 + b] = c == 0 ? w[0] : -w[c];
    }
}

// Fixed-point scale of the enumeration stage: the largest r in [0,24] such that
// len2 * max|round(entry * 2^r)| <= 2^sum_bits - 1.  Then any prefix sum is below
// 2^sum_bits in magnitude and the two halves f and g of a candidate score (each a
// sum or difference of two prefix sums) below 2^(sum_bits + 1).  sum_bits = 29 keeps f, g
// and f + g inside int32; sum_bits = 23 keeps them integers of at most 24 bits, which
// float32 holds and adds exactly.
// What the choice needs to know about a table, per candidate scale r: computed once per table
// (676 roundings per scale), so that the choice per Seq2 is a loop over 25 numbers.
void frac_table_scan(const float table[PSA_TABLE_SIZE], FracScan *fs) {
  for (int r = 0; r <= 24; ++r) {
    int64_t mx = 0;
    bool ok = true, exact = true;
    int min_tz = 63;
    for (int i = 0; i < PSA_TABLE_SIZE && ok; ++i) {
      double v = std::nearbyint(std::ldexp(double(table[i]), r));
      if (!(std::fabs(v) < 4e18)) {
        ok = false;
        break;
      }
      if (std::ldexp(v, -r) != double(table[i])) exact = false;
      const int64_t iv = std::llabs((long long)v);
      mx = std::max<int64_t>(mx, iv);
      if (iv) min_tz = std::min(min_tz, __builtin_ctzll((unsigned long long)iv));
    }
    fs->ok[r] = ok;
    fs->mx[r] = mx;
    fs->exact[r] = ok && exact;
    fs->min_tz[r] = min_tz;
  }
}

int choose_frac_bits(const FracScan &fs, int64_t len2, int sum_bits, int *exact_level) {
  const int64_t limit = (int64_t(1) << sum_bits) - 1;
  int r = 24;
  for (; r >= 0; --r)
    if (fs.ok[r] && fs.mx[r] <= limit / std::max<int64_t>(len2, 1)) break;
  if (r < 0) {
    *exact_level = -1;
    return -1;
  }
  // exactness: 1 if every entry is k * 2^-r, 2 if moreover all partial sums of up
  // to len2 entries fit the 24-bit significand of float32 at the coarsest such scale
  *exact_level = 0;
  if (fs.exact[r]) {
    *exact_level = 1;
    const int64_t mx = fs.mx[r];
    const int shift = mx == 0 ? 0 : std::min(fs.min_tz[r], r);
    if ((mx >> shift) * len2 < (int64_t(1) << 24)) *exact_level = 2;
  }
  return r;
}

int choose_frac_bits(const float table[PSA_TABLE_SIZE], int64_t len2, int sum_bits, int *exact_level) {
  FracScan fs;
  frac_table_scan(table, &fs);
  return choose_frac_bits(fs, len2, sum_bits, exact_level);
}

// Row n of the mutant triangle starts at 1-based number (n-1)*L - n(n-1)/2 + 1.
static inline int64_t row_start(int64_t n, int64_t L) { return (n - 1) * L - n * (n - 1) / 2 + 1; }

void nk_of_mutant(int64_t t, int len2, int *n, int *k) {
  const int64_t L = len2;
  const int64_t M = L * (L - 1) / 2;
  if (t < 1) t = 1;
  if (t > M) {
    // the reference's loop runs off the last row and reports n = len2 (P/cFunctions.c:108-118)
    *n = len2;
    *k = (int)(len2 + (t - M));
    return;
  }
  // solve row_start(n) <= t by the quadratic formula, then fix the rounding
  double disc = (2.0 * L - 1.0) * (2.0 * L - 1.0) - 8.0 * double(t - 1);
  int64_t row = (int64_t)(((2.0 * L - 1.0) - std::sqrt(std::max(disc, 0.0))) / 2.0) + 1;
  row = std::clamp<int64_t>(row, 1, L - 1);
  while (row > 1 && row_start(row, L) > t) --row;
  while (row < L - 1 && row_start(row + 1, L) <= t) ++row;
  *n = (int)row;
  *k = (int)(row + (t - row_start(row, L) + 1));
}

int64_t mutant_of_nk(int n, int k, int len2) { return row_start(n, len2) + (k - n) - 1; }

// Predicted cost of one Seq2, in cycles of one SIMD.
//
// Enumeration stage: the candidate loop works in tiles of 32 q x 32 p (one matrix instruction
// and 8 v_max3_f32: about 40 cycles measured with operand building, loads and hazard padding).  Per offset and level (256 positions p) it walks the row of g from the level's
// first position to the row's end: the level's own 8 blocks of q meet 1, 2, ... 8 blocks of p
// (the last of each along the diagonal, where half of the tile is not a candidate), every later
// block of q meets all 8; a row is padded to a multiple of 64 positions, of which the last 32 are
// skipped when they hold no position of Seq2.  About 1300 cycles per (offset, level) go into the
// operands of p, the requests for the rows and the lower efficiency of the level's own 36 tiles
// (fitted to uniform batches of Seq2 of 256 to 8000 characters).  Tiles per candidate therefore fall with the Seq2 length --
// a split by raw candidate counts gives the rank with the short Seq2 more time than work.
// Resolve stage alone: one dependent float32 chain step per (offset, position).
double cost_estimate(int64_t len2, int64_t len1, uint32_t stages) {
  const int64_t S = len1 - len2 + 3;
  if (S <= 0 || len2 < 2) return 0.0;
  if (stages == 0) stages = PSA_STAGE_ENUMERATE | PSA_STAGE_RESOLVE;
  if (!(stages & PSA_STAGE_ENUMERATE)) return double(S) * double(len2);
  const int64_t nruns = (len2 + 63) / 64, qend = 64 * nruns;
  int64_t tiles = 0, levels = 0;
  for (int64_t q00 = 0; q00 < qend; q00 += 256, ++levels)
    for (int64_t q = q00; q < qend; q += 64) {
      const int64_t rb = (q - q00) / 32;  // index of the first of the two blocks of q inside the level
      tiles += std::min<int64_t>(rb + 1, 8);
      if (q + 32 < len2) tiles += std::min<int64_t>(rb + 2, 8);
    }
  return double(S) * (40.0 * double(tiles) + 1300.0 * double(levels));
}

// Longest-processing-time assignment on the predicted cost of each Seq2.
int partition_by_cost(int n, const int32_t *len2, int64_t len1, int nparts, uint32_t stages, int32_t *owner) {
  if (n < 0 || nparts < 1 || (n > 0 && (!len2 || !owner))) return PSA_ERR_ARG;
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::vector<double> cost(n);
  for (int i = 0; i < n; ++i) cost[i] = cost_estimate(len2[i], len1, stages);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
  std::vector<double> load(nparts, 0.0);
  for (int idx : order) {
    int best = 0;
    for (int p = 1; p < nparts; ++p)
      if (load[p] < load[best]) best = p;
    owner[idx] = best;
    load[best] += cost[idx];
  }
  return PSA_OK;
}

}  // namespace psa

extern "C" {

void psa_create_comp_matrix(float *arr, int size, const float *weights) {
  if (!arr || !weights || size < PSA_TABLE_SIZE) return;
  psa::build_table_from_weights(weights, arr);
}

int psa_residue_class(char a, char b) { return psa::residue_class_code(a - 'A', b - 'A'); }

void psa_get_nk(int64_t mutant_number, int len2, int *n, int *k) {
  int nn = 0, kk = 0;
  psa::nk_of_mutant(mutant_number, len2, &nn, &kk);
  if (n) *n = nn;
  if (k) *k = kk;
}

int64_t psa_mutant_number(int n, int k, int len2) { return psa::mutant_of_nk(n, k, len2); }

char *psa_format_result(const psa_result *r) {
  if (!r) return nullptr;
  char *line = nullptr;
  // same conversion specifications as P/cFunctions.c:157 (its trailing "%c" of '\0' adds no visible byte)
  if (asprintf(&line, "mutant num: %d, MS(%d,%d), score: %1.2f, offset: %d\n", r->mutant_num, r->n, r->k,
               (double)r->score, r->offset) < 0)
    return nullptr;
  return line;
}

int psa_partition(int n, const int32_t *len2, int64_t len1, int nparts, int32_t *owner) {
  return psa::partition_by_cost(n, len2, len1, nparts, 0u, owner);
}

int psa_partition_stages(int n, const int32_t *len2, int64_t len1, int nparts, uint32_t stages, int32_t *owner) {
  return psa::partition_by_cost(n, len2, len1, nparts, stages, owner);
}

double psa_cost_estimate(int32_t len2, int64_t len1, uint32_t stages) { return psa::cost_estimate(len2, len1, stages); }

void psa_calc_portion(int *portion, int num) {
  // P/cFunctions.c:49-52: floor(num/2) for the first worker, the rest for the second
  if (!portion) return;
  portion[0] = num / 2;
  portion[1] = num - num / 2;
}

const char *psa_strerror(int status) {
  switch (status) {
    case PSA_OK: return "ok";
    case PSA_ERR_ARG: return "invalid argument";
    case PSA_ERR_SEQ2_SHORT: return "Seq2 shorter than 2 characters";
    case PSA_ERR_SEQ2_LONG: return "Seq2 too long for Seq1 or for the kernels";
    case PSA_ERR_RESIDUE: return "character outside 'A'..'Z'";
    case PSA_ERR_HIP: return "HIP runtime error";
    case PSA_ERR_NO_DEVICE: return "no usable GPU";
    case PSA_ERR_STATE: return "call out of order (Seq1 or table not set, or replaced under live batches; batch without context; wait without enqueue)";
    case PSA_ERR_NOMEM: return "out of memory";
    case PSA_ERR_COMM: return "the caller's all-gather failed";
    case PSA_ERR_LIMIT: return "beyond the fixed capacity of the entry point that allocates nothing";
    case PSA_ERR_CHECK: return "the enumeration stage and the resolve stage disagree";
    default: return "unknown status";
  }
}

const char *psa_version(void) { return "psa-hip 0.3 (gfx950)"; }

int psa_combine_results(const psa_result *parts, int nparts, psa_result *out) {
  if (!parts || !out || nparts < 1) return PSA_ERR_ARG;
  int bi = -1, be = -1;
  int64_t cand = 0;
  // the weakest cross-check of the shards that scored offsets: 2 (every shard's replayed shortlist gave its
  // float32 answer, so the best of those is the combined answer), 1 (bound only), 0 (some shard ran one
  // stage), negative (some shard failed)
  int check = 2;
  for (int i = 0; i < nparts; ++i) {
    const psa_result &p = parts[i];
    if (p.status != PSA_OK) {
      *out = p;
      return p.status;
    }
    cand += p.candidates;
    if (p.candidates > 0) check = (check < 0 || p.enum_check < 0) ? std::min(std::min(check, (int)p.enum_check), -1) : std::min(check, (int)p.enum_check);
    if (p.n > 0) {
      bool better = bi < 0 || p.score > parts[bi].score;
      if (!better && p.score == parts[bi].score) {
        const psa_result &q = parts[bi];
        better = p.n < q.n || (p.n == q.n && (p.k < q.k || (p.k == q.k && p.offset < q.offset)));
      }
      if (better) bi = i;
    }
    if (p.enum_frac_bits >= 0 && p.enum_n > 0) {
      bool better = be < 0 || p.enum_score > parts[be].enum_score;
      if (!better && p.enum_score == parts[be].enum_score) {
        const psa_result &q = parts[be];
        better = p.enum_n < q.enum_n ||
                 (p.enum_n == q.enum_n && (p.enum_k < q.enum_k || (p.enum_k == q.enum_k && p.enum_offset < q.enum_offset)));
      }
      if (better) be = i;
    }
  }
  *out = parts[bi >= 0 ? bi : 0];
  if (be >= 0) {
    const psa_result &e = parts[be];
    out->enum_n = e.enum_n;
    out->enum_k = e.enum_k;
    out->enum_offset = e.enum_offset;
    out->enum_score = e.enum_score;
    out->enum_frac_bits = e.enum_frac_bits;
    out->enum_exact = e.enum_exact;
    std::memcpy(out->enum_counts, e.enum_counts, sizeof e.enum_counts);
  }
  out->candidates = cand;
  out->enum_check = check;  // (maxima of pairs that agree within b agree within b)
  return PSA_OK;
}

}  // extern "C"
