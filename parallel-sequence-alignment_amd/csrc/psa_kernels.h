// Kernel parameter blocks and launch entry points (psa_kernels.hip <-> psa_api.hip).
#ifndef PSA_KERNELS_H
#define PSA_KERNELS_H

#include <hip/hip_runtime.h>

#include "psa_internal.h"

namespace psa {

constexpr int kEnumMaxNs = 64;   // most offsets in one enumeration chunk
constexpr int kEnumClasses = 6;  // (threads, registers) variants of the enumeration kernels

struct EnumParams {
  const unsigned char *seq1;       // residue codes 0..25
  int64_t len1;
  const unsigned char *seq2_pool;  // residue codes of every Seq2 of the batch
  const SeqDesc *desc;
  const Chunk *chunks;             // chunks of this kernel class
  int nchunks;
  unsigned *counter;               // work counter of this class (zeroed by psa_init_acc)
  SeqAcc *acc;
  const float *table_t;            // float table [seq2 code][seq1 code], stride 32, 27 rows
  unsigned long long *chunk_best;  // per chunk: its (maximum, smallest p) key, written by psa_enum_main
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
};

struct FinalParams {
  const unsigned char *seq1;
  const unsigned char *seq2_pool;
  const SeqDesc *desc;
  const SeqAcc *acc;
  const unsigned char *class_t;  // residue class [seq2 code][seq1 code], stride 32
  psa_result *out;
  uint32_t stages;
};

int enum_class_of(int len2);          // -1 if len2 is too long
int enum_class_positions(int c);      // positions (= threads x registers) the class covers
int enum_class_lds_bytes(int c);
int enum_class_threads(int c);

hipError_t debug_sync(const char *what);
hipError_t launch_init(SeqAcc *acc, int n, unsigned *counters, int ncounters, hipStream_t st);
hipError_t launch_enum(int cls, const EnumParams &P, int grid_main, int grid_locate, hipStream_t st,
                       hipEvent_t ev_begin, hipEvent_t ev_end);
hipError_t launch_resolve(const ResolveParams &P, int nfwd_blocks, int max_len2, int tie_grid, hipStream_t st);
hipError_t launch_finalize(const FinalParams &P, int n, hipStream_t st);

}  // namespace psa

#endif
