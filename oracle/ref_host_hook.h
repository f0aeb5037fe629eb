/*
 * ref_host_hook.h -- force-included (g++ -include) ahead of the reference's
 * "Parallel Solution/main.c" and "cFunctions.c" when oracle/Makefile builds
 * oracle/_ref/refhost_psa: the reference's own MPI host program, compiled where it lies
 * and unmodified, with its CUDA entry point bound to libpsa_hip.  This is the binding a
 * maintainer of the reference would add to P/myProto.h (the reference's sources are
 * compiled as C++, P/Makefile:2-3, so calc_best_score_CUDA has C++ linkage and takes
 * non-const pointers, P/myProto.h:20); nothing else of the host program changes.
 * TEST INFRASTRUCTURE ONLY: the test runs the binary under mpiexec -np 2 on the
 * reference's sample inputs and compares its output.txt with the golden lines.
 */
#ifndef PSA_REF_HOST_HOOK_H
#define PSA_REF_HOST_HOOK_H
extern "C" char *psa_calc_best_score(const char *seq1, const char *seq2, const float *comp_matrix);
static inline char *psa_ref_calc_best_score(char *seq1, char *seq2, float *comp_matrix) {
  return psa_calc_best_score(seq1, seq2, comp_matrix);
}
#define calc_best_score_CUDA psa_ref_calc_best_score
#endif
