/*
 * psa.h -- C ABI of libpsa_hip: the MI355X (gfx950) replacement for the
 * alignment-scoring call of ybiblow/Parallel-Sequence-Alignment.
 *
 * The reference scores, for one Seq2 against Seq1, every offset s and every
 * mutant MS(n,k) (Seq2 with the characters at 1-based positions n < k removed)
 * as the float32 left-to-right sum of 26x26 table entries and reports the first
 * maximum in (n, k, offset) order.  This library returns the same n, k, offset
 * and the same float32 bits.
 *
 * Citations: P/ = "Parallel Solution/", S/ = "Sequential Solution/" of the
 * reference repository.
 *
 * All functions return PSA_OK (0) or a psa_status code unless stated otherwise;
 * none of them exits the process except the drop-in shim psa_calc_best_score,
 * which keeps the reference's print-and-exit behaviour.
 */
#ifndef PSA_H
#define PSA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSA_TABLE_SIZE 676   /* 26 x 26, row = Seq1 residue, column = Seq2 residue (P/myProto.h:13) */
#define PSA_MAX_SEQ2 8191    /* longest Seq2 the kernels are built for */

typedef enum psa_status {
  PSA_OK = 0,
  PSA_ERR_ARG = 1,        /* NULL pointer, negative count, empty range */
  PSA_ERR_SEQ2_SHORT = 2, /* len(Seq2) < 2: no mutant exists */
  PSA_ERR_SEQ2_LONG = 3,  /* len(Seq2) - 2 > len(Seq1) (no offset) or len(Seq2) > PSA_MAX_SEQ2 */
  PSA_ERR_RESIDUE = 4,    /* a character outside 'A'..'Z' (the reference indexes out of bounds, P/cudaFunctions.cu:218) */
  PSA_ERR_HIP = 5,        /* a HIP runtime call failed; psa_last_error_text() has the message */
  PSA_ERR_NO_DEVICE = 6,  /* no usable GPU */
  PSA_ERR_STATE = 7,      /* call out of order: Seq1 or the table not set, Seq1 or the table replaced while the
                             context has batches, a batch used after its context was shut down, wait without enqueue */
  PSA_ERR_NOMEM = 8,
  PSA_ERR_COMM = 9,       /* the caller's all-gather callback reported a failure */
  PSA_ERR_LIMIT = 10,     /* beyond the fixed capacity of psa_score_range_noalloc */
  PSA_ERR_CHECK = 11      /* psa_score_batch_answer with both stages: the two stages disagree on some Seq2
                             (its record has enum_check < 0), i.e. a defect in one of them */
} psa_status;

/* stages of one run; 0 means both */
#define PSA_STAGE_ENUMERATE 1u /* score every (offset, mutant) candidate in fixed point */
#define PSA_STAGE_RESOLVE 2u   /* float32-exact answer of the reference */

/* One record per Seq2.  (n, k, offset, score) are what the reference prints
 * (P/cFunctions.c:157, S/main.c:82); counts are the residue-class counts
 * (identical, conservative, semi-conservative, other) of that winner.
 * The enum_* fields describe the enumeration stage: every candidate is scored
 * as the exact integer sum of table entries scaled by 2^enum_frac_bits and
 * rounded to nearest (the scale is the largest that keeps every such sum within
 * 24 bits, so that the kernels can add them exactly in float32; 30 bits and
 * int32 arithmetic with PSA_ENUM_I32=1 in the environment of psa_init);
 * enum_exact is 2 when that equals the float32 arithmetic
 * of the reference for every candidate (then both winners coincide), 1 when
 * only the table is represented exactly, 0 otherwise, -1 if the stage did not
 * run. */
typedef struct psa_result {
  int32_t n, k;
  int32_t offset;
  int32_t mutant_num;   /* 0-based mutant number as printed by P/cFunctions.c:153-157 */
  float score;
  uint32_t score_bits;
  int32_t counts[4];
  int32_t enum_n, enum_k, enum_offset;
  int32_t enum_score;     /* fixed point, units of 2^-enum_frac_bits */
  int32_t enum_frac_bits; /* -1: table too large for the fixed-point range */
  int32_t enum_exact;
  int32_t enum_counts[4];
  int32_t status;         /* psa_status of this Seq2 */
  int32_t enum_check;     /* the library's own cross-check of the two stages, see below: 2 or 1 agree, negative disagree, 0 not both run */
  int64_t candidates;     /* (offsets in range) x len2 x (len2-1)/2 */
} psa_result;

/* enum_check: when a run had both stages, psa_batch_wait compares the two independently computed
 * maxima of every Seq2 -- the float32 answer `score` (reference arithmetic, P/cudaFunctions.cu:210-220)
 * and the fixed-point maximum over all enumerated candidates `enum_score` * 2^-enum_frac_bits -- against
 * the rigorous rounding bound between them: with m = len2 - 2 table entries per candidate, u = 2^-24,
 * b = (m-1)u / (1 - (m-1)u) * m * max|table entry| + m * 2^-(enum_frac_bits + 1).  They must lie within b
 * of each other; -1 means they do not, i.e. a defect in one of the two stages.  No counterpart in the
 * reference, which scores once.
 *
 * The shortlist.  The candidate the reference reports has a fixed-point score within 2b of the enumerated
 * maximum, so an enumerating run lists every row (offset, n) that holds a candidate inside that window, with
 * the 256-wide blocks of k that hold one (per Seq2 up to 1024 rows and 256 listed n per offset; the run's
 * list has room for every Seq2's full quota, so whether a Seq2's list overflows depends on that Seq2 alone --
 * in a batch of more than 16384 Seq2 the quota is 2^24 rows divided by their number), and replays the listed
 * blocks in float32 exactly as calcMutantScore adds a candidate
 * (P/cudaFunctions.cu:210-220); the replayed winner -- highest score, ties to the smallest (n, k, offset) --
 * is then the reference's answer, obtained from the enumerated candidates alone.
 *   - Run with both stages: enum_check = 2 when that winner equals the resolve stage's record bit for bit
 *     (score bits, n, k, offset), -2 when it does not (a defect), and stays 1 when the list overflowed
 *     (low-entropy inputs can put every row inside the window) -- the bound above still holds.
 *   - Run with PSA_STAGE_ENUMERATE alone: n, k, offset, mutant_num, score, score_bits and counts are filled
 *     from the replayed winner when the list is complete, and stay 0 when it overflowed (run the resolve
 *     stage then); psa_batch_shortlist tells which.
 * PSA_SHORTLIST=0 in the environment of psa_init switches the shortlist off. */
typedef struct psa_ctx psa_ctx;     /* one per GPU; holds Seq1 and the table resident in HBM */
typedef struct psa_batch psa_batch; /* a set of Seq2 resident in HBM, plus work lists and result buffers */

/* ---- context ------------------------------------------------------------------
 * Replaces the implicit device 0 and the per-call cudaMalloc/cudaMemcpy of Seq1
 * and the table in calc_best_score_CUDA (P/cudaFunctions.cu:273-288). */
int psa_device_count(void);
/* The GPU a process started by a launcher should use: PSA_DEVICE if set, else the process's rank
 * on its node as the launcher exports it (OMPI_COMM_WORLD_LOCAL_RANK, MV2_COMM_WORLD_LOCAL_RANK,
 * MPI_LOCALRANKID, PMI_LOCAL_RANK, SLURM_LOCALID, LOCAL_RANK) modulo psa_device_count(), else 0.
 * The reference never selects a device: both of its ranks use device 0 of their own host
 * (P/cudaFunctions.cu:260-323, P/myHosts). */
int psa_default_device(void);
int psa_init(int device, psa_ctx **out);
/* Batches of the context that are still alive are detached: their runs are awaited and their
 * buffers released; afterwards only psa_batch_destroy may be called on them.  A context and its
 * batches are used by one thread at a time. */
void psa_shutdown(psa_ctx *ctx);
/* psa_set_seq1 / psa_set_weights / psa_set_table return PSA_ERR_STATE while the context has
 * batches: a batch is laid out for the Seq1 and the table it was created under.  Destroy the
 * batches first. */
int psa_set_seq1(psa_ctx *ctx, const char *seq1, int64_t len1);
/* table built from four weights exactly as createCompMatrix does (P/cFunctions.c:55-76) */
int psa_set_weights(psa_ctx *ctx, const float weights[4]);
/* any 26x26 float table, the argument calc_best_score_CUDA receives (P/myProto.h:20).  Every entry
 * must be finite: a NaN or an infinity returns PSA_ERR_ARG (the kernels' maxima are built on the
 * premise that no NaN can arise; the reference's plain adds would carry one through to the score). */
int psa_set_table(psa_ctx *ctx, const float table[PSA_TABLE_SIZE]);

/* Measurement aid, no counterpart in the reference.  While enabled, the candidate loops of the
 * context's runs are launched in a variant whose waves add the shader cycles and the 100 MHz ticks
 * they lived to two counters; the call waits for the GPU, returns the clock the chip held over the
 * runs since the last call (cycles / ticks x 100 MHz; 0 if no such run) and the mean wave cycles
 * per unit of work, clears the counters and sets the mode for the runs that follow.  Float32
 * matrix-core candidate loop only (the default); timings taken with the probe on are not quoted. */
int psa_clock_probe(psa_ctx *ctx, int enable, double *clock_mhz, double *wave_cycles_per_unit);

/* ---- batches of Seq2 ------------------------------------------------------------
 * Replaces the per-Seq2 loop around calc_best_score_CUDA (P/main.c:79-84, :128-135).
 * len2 may be NULL (then every seq2[i] is NUL-terminated). */
int psa_batch_create(psa_ctx *ctx, int n, const char *const *seq2, const int32_t *len2, psa_batch **out);
/* one Seq2 restricted to offsets [s_begin, s_end): the shard of a long pair that one GPU owns.
 * The range is clipped to the pair's offsets [0, len1 - len2 + 3); a range that is empty after
 * clipping (more shards than offsets) yields a valid record without a candidate: status PSA_OK,
 * n = k = 0, candidates = 0. */
int psa_batch_create_range(psa_ctx *ctx, const char *seq2, int32_t len2, int64_t s_begin, int64_t s_end,
                           psa_batch **out);
/* launches the kernels on data already resident in HBM and waits for them */
int psa_batch_run(psa_batch *b, uint32_t stages);
/* the two halves of psa_batch_run: enqueue puts one run on the context's streams and returns
 * at once; wait blocks until the records of the most recently enqueued run are in host memory.
 * Runs enqueued back to back (on this batch or on other batches of the context) execute in
 * order, so a caller that streams many batches keeps the GPU busy while it prepares the next
 * one -- the per-Seq2 blocking call of the reference (P/main.c:81, :130) has no such overlap.
 * A batch holds one set of records: enqueueing it again before wait overwrites them.
 * Enqueue returns at once unless the context already has some 500 slab passes (a few thousand kernel
 * launches: a list of thousands of Seq2) enqueued and unfinished; then it first waits for the oldest.
 * Wait for results with psa_batch_wait (the host waits for an event), not by making a stream of
 * your own wait on the library's work: the library's streams all have the default priority, and
 * a wait in a higher-priority queue on a dispatch that sits in a lower-priority queue is never
 * satisfied on this hardware (the higher queue is served first, barrier packet and all). */
int psa_batch_enqueue(psa_batch *b, uint32_t stages);
int psa_batch_wait(psa_batch *b);
/* copies the n records of the last run */
int psa_batch_results(const psa_batch *b, psa_result *out);
/* device time of the last run in milliseconds, from HIP events on the launch streams:
 * the time during which a candidate-loop kernel of the enumeration stage was executing (passes of a
 * large batch overlap), and first kernel start to last kernel end */
int psa_batch_timing(const psa_batch *b, float *enum_kernel_ms, float *total_ms);
/* Diagnostic (no counterpart in the reference): where the last run of `b` sat on the GPU's time axis, in
 * milliseconds after the start of the first kernel of the last run of `ref` (a batch of the same context that has
 * run too; b itself for times relative to its own start).  ms[0] first enumeration kernel starts, [1] first
 * candidate loop starts, [2] last candidate loop ends, [3] enumeration records written, [4] resolve chain starts,
 * [5] resolve records written; entries of stages the run did not have are -1.  Batches enqueued one after the
 * other and queried this way give the unprofiled timeline of a pipeline (tools/pipeline_timeline.py). */
int psa_batch_event_times(const psa_batch *b, const psa_batch *ref, float ms[6]);
/* per Seq2 of the last run: rows of its shortlist (see enum_check) and the list's state -- 0 no shortlist
 * (stage not run, switched off, invalid Seq2), 1 complete and replayed, 2 overflowed.  Either array may be
 * NULL.  No counterpart in the reference. */
int psa_batch_shortlist(const psa_batch *b, int32_t *count, int32_t *state);
int psa_batch_size(const psa_batch *b);
/* candidate-loop launches of one enumerating run of this batch: its rows of g are cut into passes that
 * fit one slab (256 MiB) each.  Measurement aid, no counterpart in the reference. */
int psa_batch_passes(const psa_batch *b);
void psa_batch_destroy(psa_batch *b);

/* create + run + results + destroy */
int psa_score_batch(psa_ctx *ctx, int n, const char *const *seq2, const int32_t *len2, uint32_t stages,
                    psa_result *out);
int psa_score_range(psa_ctx *ctx, const char *seq2, int32_t len2, int64_t s_begin, int64_t s_end,
                    uint32_t stages, psa_result *out);
/* psa_score_batch for a host program that lets its user choose where the answer comes from -- the
 * PSA_STAGES switch of the drop-in shim and of the host programs.  In the reference every call through
 * the boundary runs the per-candidate kernel (P/cudaFunctions.cu:290-296, called at P/main.c:81, :130).
 *   PSA_STAGE_RESOLVE                as psa_score_batch: the float32 recurrence alone;
 *   PSA_STAGE_ENUMERATE | _RESOLVE   both stages; returns PSA_ERR_CHECK (records filled) if some record has
 *                                    enum_check < 0;
 *   PSA_STAGE_ENUMERATE              every candidate is scored by the candidate loop and the answer is the
 *                                    float32 replay of the candidates near the enumerated maximum; the Seq2
 *                                    whose list overflowed (low-entropy inputs) or whose table does not fit
 *                                    the fixed-point range are then scored by the resolve stage, so that every
 *                                    valid Seq2 gets the reference's answer either way.
 * *fallbacks (may be NULL) receives the number of Seq2 that took that second path. */
int psa_score_batch_answer(psa_ctx *ctx, int n, const char *const *seq2, const int32_t *len2, uint32_t stages,
                           psa_result *out, int *fallbacks);
/* "resolve", "enumerate", "both" (what PSA_STAGES and the host programs' --stages take) -> PSA_STAGE_* bits;
 * NULL or "" -> PSA_STAGE_RESOLVE; anything else returns PSA_ERR_ARG */
int psa_parse_stages(const char *text, uint32_t *stages);

/* max-loc combine of the records that the shards of one long pair returned:
 * highest float32 score, ties to the smallest (n, k, offset); likewise for the
 * enumeration fields.  Pure host code; the transport between GPUs is the
 * caller's (one all-gather of sizeof(psa_result) bytes per rank). */
int psa_combine_results(const psa_result *parts, int nparts, psa_result *out);

/* All-gather supplied by the caller: every rank contributes `bytes` bytes at `send` and receives
 * the contributions of all ranks, in rank order, at `recv` (nranks * bytes); both are host
 * memory.  Returns 0 on success.  One line around MPI_Allgather or ncclAllGather.  `bytes` differs
 * between the calls of one run (see below) and is the same on every rank in each of them. */
typedef int (*psa_allgather_fn)(const void *send, void *recv, size_t bytes, void *comm);
#define PSA_EXCHANGE_MAX_BYTES 16  /* first exchange of an enumerating run: the shard's enumerated maximum */
#define PSA_EXCHANGE_RECORD_BYTES 136 /* last exchange: the shard's record and its replayed shortlist winner */
/* One long pair over nranks ranks, one GPU each, collectives included: cuts the pair's offsets
 * [0, len1 - len2 + 3) into nranks contiguous shards, scores shard `rank` on this context's GPU and
 * combines the shards' answers so that every rank returns the same record -- the answer
 * calc_best_score_CUDA would give for the whole pair (P/myProto.h:20; called per Seq2 at
 * P/main.c:81, :130).  With nranks > 1 a run calls `allgather(.., comm)`
 *   - once with PSA_EXCHANGE_MAX_BYTES per rank when the stages include the enumeration, after the
 *     candidate loops: the shards' enumerated maxima, so that every shard shortlists against the GLOBAL
 *     maximum (a shard far below it lists nothing) and the replayed candidates of all shards together
 *     hold the reference's answer, as they do on one GPU;
 *   - once with PSA_EXCHANGE_RECORD_BYTES per rank at the end: record and replayed winner of every shard,
 *     combined by the max-loc rule of psa_combine_results; enum_check of the combined record is 2 when the
 *     best replayed winner of all shards equals the combined float32 answer bit for bit (-2 if it does
 *     not), 1 when some shard's list overflowed.
 * Every rank of the communicator must call it, with the same stages; with nranks == 1 the callback may
 * be NULL (no exchange; if one is given, the run goes through it as a one-rank all-gather).  A rank whose scoring fails, or whose context is not ready, still takes part in every
 * all-gather of the run and all ranks return that status.  A run of PSA_STAGE_ENUMERATE alone returns the
 * float32 fields of the best replayed winner, or zeros if some shard's list overflowed. */
int psa_score_range_allgather(psa_ctx *ctx, const char *seq2, int32_t len2, int rank, int nranks,
                              psa_allgather_fn allgather, void *comm, uint32_t stages, psa_result *out);
/* The same on a resident batch: `shard` was made by psa_batch_create_range for this rank's offsets
 * (shard r of nranks owns [S*r/nranks, S*(r+1)/nranks), S = len1 - len2 + 3); one run, the all-gathers
 * described above, the combine.  For callers that score the same pair repeatedly or keep their
 * shard in HBM. */
int psa_batch_run_allgather(psa_batch *shard, int nranks, psa_allgather_fn allgather, void *comm,
                            uint32_t stages, psa_result *out);

/* ---- one entry point that allocates nothing --------------------------------------------
 * No counterpart in the reference (whose calc_best_score_CUDA allocates five device buffers per
 * call and frees none, P/cudaFunctions.cu:273-288).  For callers that cannot allocate: code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 *
 * psa_score_range_noalloc is psa_score_range with the resolve stage (the reference's float32
 * answer: n, k, offset, score bits, class counts), inputs and output already in device memory:
 * d_seq1 and d_seq2 hold the characters 'A'..'Z' (no terminator needed), d_table the 26 x 26
 * float table calc_best_score_CUDA receives, and the record is written to d_out (device memory,
 * or pinned host memory the device can write).  The kernels are enqueued on `stream` (a
 * hipStream_t; NULL = the default stream) of the current device and the call returns without
 * waiting: the record is complete when the stream has run that far.  Every intermediate lives in
 * one fixed array of PSA_NOALLOC_SCRATCH_BYTES (3.3 MiB) of device memory inside the library's
 * code object, which every GPU the library is loaded on holds until the process ends.  The call
 * makes no hipMalloc, hipFree or other allocating call.
 * Limits: len1 <= PSA_NOALLOC_MAX_SEQ1 (else PSA_ERR_LIMIT), len2 <= PSA_MAX_SEQ2.  The
 * enumeration fields of the record are cleared (enum_frac_bits = -1).  A character outside
 * 'A'..'Z' is found on the device: the record then has status PSA_ERR_RESIDUE and nothing else;
 * likewise a table entry that is not finite (the requirement of psa_set_table): PSA_ERR_ARG.
 * All calls on one GPU share that GPU's array: calls whose kernels could overlap (two streams,
 * two threads) race; issue them on one stream or order them with events.
 * psa_noalloc_init (optional; otherwise done by the first call on each device) looks up the
 * array's address on the current device and sets the kernels' LDS limits there -- call it before
 * a graph capture begins. */
#define PSA_NOALLOC_MAX_SEQ1 262144
#define PSA_NOALLOC_SCRATCH_BYTES 3430912
int psa_noalloc_init(void);
int psa_score_range_noalloc(const char *d_seq1, int64_t len1, const char *d_seq2, int32_t len2,
                            const float *d_table, int64_t s_begin, int64_t s_end, void *stream,
                            psa_result *d_out);
const char *psa_noalloc_last_error_text(void);

/* ---- the reference's host helpers ------------------------------------------------ */
/* drop-in for  char* calc_best_score_CUDA(char*, char*, float*)  (P/myProto.h:20,
 * P/cudaFunctions.cu:260-323): returns the malloc'd line
 * "mutant num: %d, MS(%d,%d), score: %1.2f, offset: %d\n" (caller frees, P/main.c:83),
 * echoes it on stdout (P/cudaFunctions.cu:321; PSA_SHIM_QUIET=1 silences it), and on
 * any failure prints to stderr and exits like the reference's macros (P/cudaFunctions.cu:5-18).
 * Keeps one context per process on GPU psa_default_device() (so the ranks that mpiexec starts
 * on one node take different GPUs); calls from several threads are serialised. */
char *psa_calc_best_score(const char *seq1, const char *seq2, const float *comp_matrix);
/* createCompMatrix (P/cFunctions.c:55-64); size must be PSA_TABLE_SIZE */
void psa_create_comp_matrix(float *arr, int size, const float *weights);
/* findSimilarityWeight's class decision (P/cFunctions.c:67-100): 0 identical, 1 conservative,
 * 2 semi-conservative, 3 other */
int psa_residue_class(char a, char b);
/* CPUGetNK (P/cFunctions.c:104-119): 1-based mutant number -> (n, k); closed form */
void psa_get_nk(int64_t mutant_number, int len2, int *n, int *k);
int64_t psa_mutant_number(int n, int k, int len2);
/* the line calcBestScoreOmp formats (P/cFunctions.c:157); malloc'd, caller frees */
char *psa_format_result(const psa_result *r);
/* work split replacing calcPortion (P/cFunctions.c:49-52): assigns each Seq2 to one of nparts
 * workers (owner[i] in [0, nparts)), largest first onto the least loaded, balancing the predicted
 * GPU time of the stages that will run (0 = both): with the enumeration stage the number of
 * 32 x 32 tiles the candidate loop works through, which per candidate falls with the Seq2 length
 * (diagonal tiles, padding and per-offset set-up weigh more on a short Seq2), not the raw
 * candidate count; with the resolve stage alone offsets x length.  psa_partition is
 * psa_partition_stages with stages = 0. */
int psa_partition(int n, const int32_t *len2, int64_t len1, int nparts, int32_t *owner);
int psa_partition_stages(int n, const int32_t *len2, int64_t len1, int nparts, uint32_t stages, int32_t *owner);
/* the cost that split balances, for one Seq2 (cycles of one SIMD; 0 for an invalid length) */
double psa_cost_estimate(int32_t len2, int64_t len1, uint32_t stages);
/* the reference's own split (first n/2 to worker 0, rest to worker 1) for comparison */
void psa_calc_portion(int *portion, int num);

const char *psa_strerror(int status);
const char *psa_last_error_text(void);
const char *psa_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PSA_H */
