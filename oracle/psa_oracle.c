/*
 * psa_oracle.c -- CPU oracle for the two-deletion mutant alignment score.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing the product ships or measures may call
 * into this file: it exists so that tests, the smoke check and the CPU
 * baseline leg of the benchmark have an independent statement of what the
 * reference computes.  Parity status: PINNED -- every function below is
 * checked against the reference's own Sequential Solution compiled from
 * /root/reference (oracle/_ref) and against the committed golden vectors
 * under tests/golden/.
 *
 * Reference citations use S/ = "Sequential Solution/", P/ = "Parallel Solution/".
 *
 * Conventions: sequences are arrays of bytes 'A'..'Z'; L1 = len(seq1),
 * L2 = len(seq2); a mutant MS(n,k), 1 <= n < k <= L2, is seq2 with the
 * characters at 0-based positions n-1 and k-1 removed; it has m = L2-2
 * characters and is laid against seq1[s .. s+m-1] for offset s in
 * [0, L1-m]; the score is the float32 left-to-right sum of table entries.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PSA_TBL 676

typedef struct {
  int32_t n, k;        /* 1-based deleted positions, n < k */
  int32_t offset;      /* winning offset */
  int32_t mutant_num;  /* 0-based mutant number as printed by P/cFunctions.c:157 */
  float score;         /* float32 score */
  uint32_t score_bits; /* the same, as raw bits */
  int32_t status;      /* 0 ok, nonzero = invalid input */
} psa_oracle_result;

/* ---- residue classes: S/main.c:135-157 (P/cFunctions.c:78-100) ------------- */

static int in_group(const char *g, int c) { return c != 0 && strchr(g, c) != NULL; }

/* 0 identical, 1 conservative, 2 semi-conservative, 3 other
 * (priority order of S/main.c:124-133). */
int psa_oracle_class(int a, int b) {
  static const char *cons[9] = {"NDEQ", "NEQK", "STA", "MILV", "QHRK", "NHQK", "FYW", "HY", "MILF"};
  static const char *semi[11] = {"SAG", "ATV", "CSA", "SGND", "STPA", "STNK",
                                 "NEQHRK", "NDEQHK", "SNDEQK", "HFY", "FVLIM"};
  int i;
  if (a == b) return 0;
  for (i = 0; i < 9; ++i)
    if (in_group(cons[i], a) && in_group(cons[i], b)) return 1;
  for (i = 0; i < 11; ++i)
    if (in_group(semi[i], a) && in_group(semi[i], b)) return 2;
  return 3;
}

/* 26x26 class digits, row = seq1 residue, column = seq2 residue. */
void psa_oracle_class_table(uint8_t *out) {
  int a, b;
  for (a = 0; a < 26; ++a)
    for (b = 0; b < 26; ++b) out[a * 26 + b] = (uint8_t)psa_oracle_class('A' + a, 'A' + b);
}

/* S/main.c:113-133: +W[0] identical, -W[1], -W[2], -W[3]. */
void psa_oracle_comp_matrix(const float *w, float *out) {
  int a, b;
  for (a = 0; a < 26; ++a)
    for (b = 0; b < 26; ++b) {
      int c = psa_oracle_class('A' + a, 'A' + b);
      out[a * 26 + b] = (c == 0) ? w[0] : -w[c];
    }
}

/* ---- mutant numbering: S/main.c:159-174 -------------------------------------
 * t is the 1-based mutant number; rows of the strict upper triangle in
 * ascending n, row n holding k = n+1..L2. */
void psa_oracle_get_nk(int64_t t, int L2, int *n, int *k) {
  int row = 1;
  int64_t row_len = (int64_t)L2 - 1;
  while (row < L2 && t > row_len) {
    t -= row_len;
    --row_len;
    ++row;
  }
  *n = row;
  *k = row + (int)t;
}

int64_t psa_oracle_mutant_number(int n, int k, int L2) {
  /* 1-based; inverse of psa_oracle_get_nk */
  return (int64_t)(n - 1) * L2 - (int64_t)n * (n - 1) / 2 + (k - n);
}

static int validate(const uint8_t *seq1, int64_t L1, const uint8_t *seq2, int64_t L2) {
  int64_t i;
  if (L2 < 2 || L1 < 1) return 1;
  if (L2 - 2 > L1) return 2;
  for (i = 0; i < L1; ++i)
    if (seq1[i] < 'A' || seq1[i] > 'Z') return 3;
  for (i = 0; i < L2; ++i)
    if (seq2[i] < 'A' || seq2[i] > 'Z') return 3;
  return 0;
}

static uint32_t fbits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}

/* ---- brute force: literal restatement of S/main.c:44-83 ---------------------
 * mutants outer (1-based numbers t in [t_begin, t_end)), offsets inner,
 * positions innermost; float32 accumulation from 0; strict '>' keeps the first
 * maximum.  The initial best is -inf rather than the reference's -100000
 * (S/main.c:49): the two agree whenever any score exceeds -100000.
 * Offsets are restricted to [s_begin, s_end) so shards can be checked. */
int psa_oracle_brute(const uint8_t *seq1, int64_t L1, const uint8_t *seq2, int64_t L2,
                     const float *table, int64_t t_begin, int64_t t_end, int64_t s_begin,
                     int64_t s_end, psa_oracle_result *out) {
  int64_t t, s, i;
  int64_t m = L2 - 2;
  volatile float best = -INFINITY;
  int best_n = 0, best_k = 0;
  int64_t best_s = -1, best_t = 0;
  memset(out, 0, sizeof *out);
  if ((out->status = validate(seq1, L1, seq2, L2)) != 0) return out->status;
  for (t = t_begin; t < t_end; ++t) {
    int n, k;
    psa_oracle_get_nk(t, (int)L2, &n, &k);
    for (s = s_begin; s < s_end; ++s) {
      float score = 0.0f;
      int64_t j = 0;
      for (i = 0; i < m; ++i, ++j) {
        if (j == n - 1) ++j;
        if (j == k - 1) ++j;
        score += table[(seq1[s + i] - 'A') * 26 + (seq2[j] - 'A')];
      }
      if (score > best) {
        best = score;
        best_n = n;
        best_k = k;
        best_s = s;
        best_t = t;
      }
    }
  }
  out->n = best_n;
  out->k = best_k;
  out->offset = (int32_t)best_s;
  out->mutant_num = (int32_t)(best_t - 1);
  out->score = best;
  out->score_bits = fbits(best);
  return 0;
}

/* ---- class counts of one candidate (exact integers) ------------------------- */
int psa_oracle_counts(const uint8_t *seq1, const uint8_t *seq2, int64_t L2, int64_t s, int n, int k,
                      int32_t *counts) {
  int64_t i, j = 0, m = L2 - 2;
  counts[0] = counts[1] = counts[2] = counts[3] = 0;
  for (i = 0; i < m; ++i, ++j) {
    if (j == n - 1) ++j;
    if (j == k - 1) ++j;
    counts[psa_oracle_class(seq1[s + i], seq2[j])]++;
  }
  return 0;
}

/* ---- fast exact path ---------------------------------------------------------
 * Rounding to nearest is monotone: x <= y implies fl(x+c) <= fl(y+c).  So the
 * float32 maximum over all (n,k) for one offset follows from a three-state
 * recurrence over seq2 positions j (0, 1 or 2 deletions so far), and the first
 * maximiser in (n, k, offset) order from backward thresholds.  D_d(s,j) is the
 * table entry that seq2[j] meets after d deletions: table[seq1[s+j-d]][seq2[j]].
 */
static inline float Dval(const uint8_t *seq1, const uint8_t *seq2, const float *table, int64_t s,
                         int64_t j, int d) {
  return table[(seq1[s + j - d] - 'A') * 26 + (seq2[j] - 'A')];
}

/* order-preserving map float32 -> uint32 and back */
static inline uint32_t f2key(float f) {
  uint32_t u = fbits(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float key2f(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* smallest float x (possibly -inf) with fl(x + d) >= t; +inf if none does. */
static float inv_add(float t, float d) {
  uint32_t lo = f2key(-INFINITY), hi = f2key(INFINITY);
  volatile float probe;
  probe = INFINITY + d;
  if (!(probe >= t)) return INFINITY;
  /* invariant: predicate true at hi; find the first key where it holds */
  while (lo < hi) {
    uint32_t mid = lo + (hi - lo) / 2;
    probe = key2f(mid) + d;
    if (probe >= t)
      hi = mid;
    else
      lo = mid + 1;
  }
  return key2f(lo);
}

/* max over (n,k) for one offset; float32, bit-exact */
static float forward_one(const uint8_t *seq1, const uint8_t *seq2, int64_t L2, const float *table,
                         int64_t s) {
  volatile float a = 0.0f, b = -INFINITY, c = -INFINITY;
  int64_t j;
  for (j = 0; j < L2; ++j) {
    float na = a, nb = b, nc = c;
    /* third state: delete position j out of state b, or consume it */
    nc = b;
    if (j >= 2 && c != -INFINITY) {
      volatile float v = c + Dval(seq1, seq2, table, s, j, 2);
      if (v > nc) nc = v;
    }
    /* second state: delete position j out of state a, or consume it */
    nb = -INFINITY;
    if (j <= L2 - 2) {
      nb = a;
      if (j >= 1 && b != -INFINITY) {
        volatile float v = b + Dval(seq1, seq2, table, s, j, 1);
        if (v > nb) nb = v;
      }
    }
    if (j <= L2 - 3) na = a + Dval(seq1, seq2, table, s, j, 0);
    a = na;
    b = nb;
    c = nc;
  }
  return c;
}

int psa_oracle_fast(const uint8_t *seq1, int64_t L1, const uint8_t *seq2, int64_t L2,
                    const float *table, int64_t s_begin, int64_t s_end, psa_oracle_result *out) {
  int64_t s, j, S;
  float best = -INFINITY;
  float *F, *tc, *tb, *av;
  int64_t bp = -1, bq = -1, bs = -1;
  memset(out, 0, sizeof *out);
  if ((out->status = validate(seq1, L1, seq2, L2)) != 0) return out->status;
  S = L1 - L2 + 3;
  if (s_begin < 0) s_begin = 0;
  if (s_end > S) s_end = S;
  if (s_end <= s_begin) return out->status = 4;
  F = (float *)malloc(sizeof(float) * (size_t)(s_end - s_begin));
  tc = (float *)malloc(sizeof(float) * (size_t)(L2 + 2));
  tb = (float *)malloc(sizeof(float) * (size_t)(L2 + 2));
  av = (float *)malloc(sizeof(float) * (size_t)(L2 + 2));
  for (s = s_begin; s < s_end; ++s) {
    F[s - s_begin] = forward_one(seq1, seq2, L2, table, s);
    if (F[s - s_begin] > best) best = F[s - s_begin];
  }
  for (s = s_begin; s < s_end; ++s) {
    int64_t p, q;
    volatile float v;
    if (F[s - s_begin] != best) continue;
    /* thresholds: tc[j] = least value of the third state before position j that
     * still ends at 'best'; tb[j] likewise for the second state. */
    tc[L2] = best;
    for (j = L2 - 1; j >= 2; --j) tc[j] = inv_add(tc[j + 1], Dval(seq1, seq2, table, s, j, 2));
    tb[L2 - 1] = tc[L2];
    for (j = L2 - 2; j >= 1; --j) {
      float cont = inv_add(tb[j + 1], Dval(seq1, seq2, table, s, j, 1));
      tb[j] = tc[j + 1] < cont ? tc[j + 1] : cont;
    }
    /* prefix values of the undeleted chain */
    v = 0.0f;
    av[0] = 0.0f;
    for (j = 0; j <= L2 - 3; ++j) {
      v = v + Dval(seq1, seq2, table, s, j, 0);
      av[j + 1] = v;
    }
    for (p = 0; p <= L2 - 2; ++p)
      if (av[p] >= tb[p + 1]) break;
    if (p > L2 - 2) continue; /* cannot happen when F[s] == best */
    v = av[p];
    for (q = p + 1; q <= L2 - 1; ++q) {
      if (v >= tc[q + 1]) break;
      v = v + Dval(seq1, seq2, table, s, q, 1);
    }
    if (bs < 0 || p < bp || (p == bp && q < bq)) { /* s ascending: ties keep the earlier s */
      bp = p;
      bq = q;
      bs = s;
    }
    /* (n,k) = (1,2) is the first mutant there is and s ascends: no later offset can come before this one
     * (without this an all-same pair of 200000 x 8000 walks the thresholds of 192003 tied offsets) */
    if (bp == 0 && bq == 1) break;
  }
  free(F);
  free(tc);
  free(tb);
  free(av);
  out->n = (int32_t)(bp + 1);
  out->k = (int32_t)(bq + 1);
  out->offset = (int32_t)bs;
  out->mutant_num = (int32_t)(psa_oracle_mutant_number(out->n, out->k, (int)L2) - 1);
  out->score = best;
  out->score_bits = fbits(best);
  return 0;
}

/* per-offset float32 maxima (for checking the device forward pass) */
int psa_oracle_forward(const uint8_t *seq1, int64_t L1, const uint8_t *seq2, int64_t L2,
                       const float *table, float *F) {
  int64_t s, S = L1 - L2 + 3;
  int st = validate(seq1, L1, seq2, L2);
  if (st) return st;
  for (s = 0; s < S; ++s) F[s] = forward_one(seq1, seq2, L2, table, s);
  return 0;
}

/* ---- fixed-point enumeration score -------------------------------------------
 * The enumerating stage scores a candidate as the exact integer sum of table
 * entries scaled by 2^r and rounded to nearest-even.  r is the largest value in
 * [0,24] with L2 * max|entry| <= 2^sum_bits - 1.  sum_bits = 29 keeps every partial
 * sum and every sum of two partial sums inside int32; sum_bits = 23 keeps them
 * integers of at most 24 bits, which float32 adds exactly (the library's default
 * arithmetic for this stage).  Returns r, or -1 if even r = 0
 * does not fit; *exact is 0 when the rounding changed some entry, 1 when it
 * changed none, and 2 when in addition L2 * max|entry| < 2^24, so that every
 * partial sum the float32 path forms is representable and the two paths must
 * agree exactly. */
int psa_oracle_fixed_table(const float *table, int64_t L2, int sum_bits, int32_t *itab, int *exact) {
  int r, i;
  for (r = 24; r >= 0; --r) {
    int64_t mx = 0;
    int ok = 1;
    for (i = 0; i < PSA_TBL; ++i) {
      double v = nearbyint(ldexp((double)table[i], r));
      int64_t iv;
      if (!(fabs(v) < 1e15)) {
        ok = 0;
        break;
      }
      iv = (int64_t)v;
      if (iv < 0) iv = -iv;
      if (iv > mx) mx = iv;
    }
    if (ok && mx * L2 <= ((int64_t)1 << sum_bits) - 1) break;
  }
  if (r < 0) return -1;
  *exact = 1;
  {
    int64_t mx = 0;
    for (i = 0; i < PSA_TBL; ++i) {
      double v = nearbyint(ldexp((double)table[i], r));
      itab[i] = (int32_t)v;
      if (ldexp(v, -r) != (double)table[i]) *exact = 0;
      if (llabs((long long)itab[i]) > mx) mx = llabs((long long)itab[i]);
    }
    if (*exact) {
      /* smallest scale r0 at which every entry is an integer */
      int r0 = r;
      while (r0 > 0) {
        int all_even = 1;
        for (i = 0; i < PSA_TBL; ++i)
          if ((itab[i] >> (r - r0)) & 1) all_even = 0;
        if (!all_even) break;
        --r0;
      }
      if ((mx >> (r - r0)) * L2 < ((int64_t)1 << 24)) *exact = 2;
    }
  }
  return r;
}

typedef struct {
  int64_t score; /* fixed-point integer score */
  int32_t n, k, offset;
  int32_t status;
} psa_oracle_enum_result;

/* every candidate scored one by one; first maximum in (n,k,offset) order */
int psa_oracle_enum_brute(const uint8_t *seq1, int64_t L1, const uint8_t *seq2, int64_t L2,
                          const int32_t *itab, int64_t s_begin, int64_t s_end,
                          psa_oracle_enum_result *out) {
  int64_t p, q, s, i, m = L2 - 2;
  int have = 0;
  memset(out, 0, sizeof *out);
  if ((out->status = validate(seq1, L1, seq2, L2)) != 0) return out->status;
  for (p = 0; p <= L2 - 2; ++p)
    for (q = p + 1; q <= L2 - 1; ++q)
      for (s = s_begin; s < s_end; ++s) {
        int64_t sc = 0, j = 0;
        for (i = 0; i < m; ++i, ++j) {
          if (j == p) ++j;
          if (j == q) ++j;
          sc += itab[(seq1[s + i] - 'A') * 26 + (seq2[j] - 'A')];
        }
        if (!have || sc > out->score) {
          have = 1;
          out->score = sc;
          out->n = (int32_t)p + 1;
          out->k = (int32_t)q + 1;
          out->offset = (int32_t)s;
        }
      }
  return 0;
}

/* same answer in O(L1*L2): per offset, score = f(p) + g(q) with
 * f(p) = sum_{j<p} D0 - sum_{1<=j<=p} D1,  g(q) = sum_{1<=j<q} D1 + sum_{j>q} D2. */
int psa_oracle_enum_fast(const uint8_t *seq1, int64_t L1, const uint8_t *seq2, int64_t L2,
                         const int32_t *itab, int64_t s_begin, int64_t s_end,
                         psa_oracle_enum_result *out) {
  int64_t s, j;
  int64_t *f, *g, *sufv;
  int32_t *sufi;
  int have = 0;
  memset(out, 0, sizeof *out);
  if ((out->status = validate(seq1, L1, seq2, L2)) != 0) return out->status;
  f = (int64_t *)malloc(sizeof(int64_t) * (size_t)(L2 + 1));
  g = (int64_t *)malloc(sizeof(int64_t) * (size_t)(L2 + 1));
  sufv = (int64_t *)malloc(sizeof(int64_t) * (size_t)(L2 + 1));
  sufi = (int32_t *)malloc(sizeof(int32_t) * (size_t)(L2 + 1));
  for (s = s_begin; s < s_end; ++s) {
    int64_t p0 = 0, p1 = 0, suf2 = 0;
    /* f[p] = P0[p] - P1[p+1] */
    for (j = 0; j <= L2 - 2; ++j) {
      if (j >= 1) p1 += itab[(seq1[s + j - 1] - 'A') * 26 + (seq2[j] - 'A')];
      f[j] = p0 - p1;
      if (j <= L2 - 3) p0 += itab[(seq1[s + j] - 'A') * 26 + (seq2[j] - 'A')];
    }
    /* g[q] = P1[q] + sum_{j>q} D2 ; build from the right */
    {
      int64_t pre1 = 0;
      for (j = 1; j <= L2 - 1; ++j) {
        g[j] = pre1; /* sum_{1<=i<j} D1 */
        if (j <= L2 - 2) pre1 += itab[(seq1[s + j - 1] - 'A') * 26 + (seq2[j] - 'A')];
      }
      for (j = L2 - 1; j >= 1; --j) {
        g[j] += suf2;
        if (j >= 2) suf2 += itab[(seq1[s + j - 2] - 'A') * 26 + (seq2[j] - 'A')];
      }
    }
    /* suffix maxima of g with the smallest index among equals */
    sufv[L2 - 1] = g[L2 - 1];
    sufi[L2 - 1] = (int32_t)(L2 - 1);
    for (j = L2 - 2; j >= 1; --j) {
      if (g[j] >= sufv[j + 1]) {
        sufv[j] = g[j];
        sufi[j] = (int32_t)j;
      } else {
        sufv[j] = sufv[j + 1];
        sufi[j] = sufi[j + 1];
      }
    }
    for (j = 0; j <= L2 - 2; ++j) {
      int64_t sc = f[j] + sufv[j + 1];
      int32_t n = (int32_t)j + 1, k = sufi[j + 1] + 1;
      int better = !have || sc > out->score ||
                   (sc == out->score && (n < out->n || (n == out->n && k < out->k)));
      if (better) {
        have = 1;
        out->score = sc;
        out->n = n;
        out->k = k;
        out->offset = (int32_t)s;
      }
    }
  }
  free(f);
  free(g);
  free(sufv);
  free(sufi);
  return 0;
}

/* fixed-point score of one named candidate */
int64_t psa_oracle_enum_score(const uint8_t *seq1, const uint8_t *seq2, int64_t L2,
                              const int32_t *itab, int64_t s, int n, int k) {
  int64_t i, j = 0, m = L2 - 2, sc = 0;
  for (i = 0; i < m; ++i, ++j) {
    if (j == n - 1) ++j;
    if (j == k - 1) ++j;
    sc += itab[(seq1[s + i] - 'A') * 26 + (seq2[j] - 'A')];
  }
  return sc;
}

/* ---- multi-core brute force for the host-CPU baseline ------------------------
 * Same arithmetic as psa_oracle_brute over mutants [t_begin, t_end), all
 * offsets; threads take contiguous mutant ranges and the partial winners are
 * merged in mutant order so the first maximum is kept. */
#ifdef _OPENMP
#include <omp.h>
#endif
int psa_oracle_brute_mt(const uint8_t *seq1, int64_t L1, const uint8_t *seq2, int64_t L2,
                        const float *table, int64_t t_begin, int64_t t_end, int nthreads,
                        psa_oracle_result *out) {
  int64_t S = L1 - L2 + 3;
  int T = nthreads < 1 ? 1 : nthreads, i;
  psa_oracle_result *part;
  memset(out, 0, sizeof *out);
  if ((out->status = validate(seq1, L1, seq2, L2)) != 0) return out->status;
  part = (psa_oracle_result *)calloc((size_t)T, sizeof *part);
#ifdef _OPENMP
#pragma omp parallel for num_threads(T) schedule(static, 1)
#endif
  for (i = 0; i < T; ++i) {
    int64_t span = t_end - t_begin;
    int64_t b = t_begin + span * i / T, e = t_begin + span * (i + 1) / T;
    psa_oracle_brute(seq1, L1, seq2, L2, table, b, e, 0, S, &part[i]);
    if (e <= b) part[i].offset = -1;
  }
  out->offset = -1;
  out->score = -INFINITY;
  for (i = 0; i < T; ++i)
    if (part[i].offset >= 0 && part[i].score > out->score) *out = part[i];
  out->score_bits = fbits(out->score);
  free(part);
  return 0;
}

int psa_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
