// psa_mpi: the reference's MPI host program (P/main.c) on top of libpsa_hip, for any number
// of ranks on any number of nodes, one rank per GPU.
//
// Protocol as in the reference: rank 0 reads "input.txt" (P/cFunctions.c:17-37), every rank
// gets the weights, Seq1 and its share of the Seq2 (P/main.c:54-76, :104-126), scores them,
// and rank 0 writes one line per Seq2 in input order to "output.txt" and echoes it
// (P/main.c:79-94).  What changes: the reference is fixed to two ranks (P/main.c:20-24) and
// splits by count (P/cFunctions.c:49-52, rank 0 takes the first half); here the split
// balances the predicted GPU time over all ranks (psa_partition_stages), the sequences travel as one
// broadcast instead of one MPI_Send per Seq2, and the answers come back as fixed-size
// records (one MPI_Gatherv) instead of formatted lines.  Buffers are dynamic.
//
//   mpiexec -np N psa_mpi [--dry-run] [--long-pair] [--stages=resolve|enumerate|both] [input.txt [output.txt]]
//
// --stages (default: PSA_STAGES in the environment, else resolve) chooses where the answer comes from:
// the float32 recurrence alone, the candidate loop over every (offset, mutant) candidate -- what every
// call of the reference runs, P/cudaFunctions.cu:290-296 -- with the float32 replay of the candidates near
// its maximum, or both with the two required to agree.
// A rank uses GPU (rank on its node) mod (GPUs of the node).  --dry-run does everything but
// the scoring and prints which rank owns which Seq2: a check of the plumbing without a GPU.
// --long-pair scores every Seq2 on all ranks together instead of giving each Seq2 to one rank:
// the offsets of the pair are cut into one contiguous shard per rank, and one MPI_Allgather of
// one record per rank inside psa_score_range_allgather brings the shards' answers together (for
// a Seq1 so long that one pair is the whole job; the output is the same).
#include <mpi.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "psa.h"
#include "psa_input.h"

// the all-gather psa_score_range_allgather calls: comm points to the MPI communicator
static int mpi_allgather(const void *send, void *recv, size_t bytes, void *comm) {
  return MPI_Allgather(send, (int)bytes, MPI_BYTE, recv, (int)bytes, MPI_BYTE, *(MPI_Comm *)comm) == MPI_SUCCESS ? 0 : 1;
}

int main(int argc, char **argv) {
  MPI_Init(&argc, &argv);
  int rank = 0, size = 1;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  bool dry = false, long_pair = false;
  const char *stages_text = std::getenv("PSA_STAGES");
  std::vector<const char *> pos;
  for (int i = 1; i < argc; ++i) {
    if (std::strcmp(argv[i], "--dry-run") == 0)
      dry = true;
    else if (std::strcmp(argv[i], "--long-pair") == 0)
      long_pair = true;
    else if (const char *w = psa_host::stages_option(argc, argv, i))
      stages_text = w;
    else
      pos.push_back(argv[i]);
  }
  uint32_t stages = PSA_STAGE_RESOLVE;
  if (psa_parse_stages(stages_text, &stages) != PSA_OK) {  // the same on every rank
    if (rank == 0) std::fprintf(stderr, "psa_mpi: %s\n", psa_last_error_text());
    MPI_Finalize();
    return 2;
  }
  const char *in_name = pos.size() > 0 ? pos[0] : "input.txt";    // P/myProto.h:8
  const char *out_name = pos.size() > 1 ? pos[1] : "output.txt";  // P/myProto.h:9
  auto t_start = std::chrono::steady_clock::now();

  // rank 0 reads; everything travels in two broadcasts (a header, then the characters)
  float weights[4] = {0, 0, 0, 0};
  std::string seq1;
  std::vector<std::string> seq2;
  long long header[3] = {0, 0, -1};  // Seq1 length, number of Seq2, status
  FILE *out = nullptr;
  if (rank == 0) {
    std::printf("my rank is %d and i'm currently reading from the file\n", rank);  // P/main.c:37
    psa_host::Input input;
    if (!psa_host::read_input(in_name, input)) {
      std::printf("Error in opening the file\n");  // P/cFunctions.c:21-24
    } else {
      std::memcpy(weights, input.weights, sizeof weights);
      seq1.swap(input.seq1);
      seq2.swap(input.seq2);
      out = dry ? nullptr : std::fopen(out_name, "w");
      if (!dry && !out) {
        std::printf("Error in opening the file\n");  // P/main.c:46-49
      } else {
        header[0] = (long long)seq1.size();
        header[1] = (long long)seq2.size();
        header[2] = 0;
        std::printf("num of seq2 = %d\n", input.count);  // P/main.c:51
      }
    }
  }
  MPI_Bcast(header, 3, MPI_LONG_LONG, 0, MPI_COMM_WORLD);
  if (header[2] != 0) {
    MPI_Finalize();
    return 1;
  }
  const int n = (int)header[1];
  MPI_Bcast(weights, 4, MPI_FLOAT, 0, MPI_COMM_WORLD);
  std::vector<int32_t> len2((size_t)n);
  if (rank == 0)
    for (int i = 0; i < n; ++i) len2[(size_t)i] = (int32_t)seq2[(size_t)i].size();
  if (n > 0) MPI_Bcast(len2.data(), n, MPI_INT32_T, 0, MPI_COMM_WORLD);
  long long total = header[0];
  for (int i = 0; i < n; ++i) total += len2[(size_t)i];
  std::vector<char> blob((size_t)total);
  if (rank == 0) {
    size_t at = 0;
    std::memcpy(blob.data(), seq1.data(), seq1.size());
    at += seq1.size();
    for (const auto &s : seq2) {
      std::memcpy(blob.data() + at, s.data(), s.size());
      at += s.size();
    }
  }
  for (long long at = 0; at < total; at += (1 << 30)) {  // MPI counts are ints
    const long long cnt = std::min<long long>(total - at, 1 << 30);
    MPI_Bcast(blob.data() + at, (int)cnt, MPI_CHAR, 0, MPI_COMM_WORLD);
  }
  seq1.assign(blob.data(), (size_t)header[0]);

  int local = 0;  // my place among the ranks of this node (a collective: every rank takes part)
  {
    MPI_Comm node;
    MPI_Comm_split_type(MPI_COMM_WORLD, MPI_COMM_TYPE_SHARED, rank, MPI_INFO_NULL, &node);
    MPI_Comm_rank(node, &local);
    MPI_Comm_free(&node);
  }

  if (long_pair && !dry) {
    // every Seq2 is scored by all ranks together, offsets sharded, one all-gather per Seq2
    const int ngpu = psa_device_count();
    psa_ctx *ctx = nullptr;
    int st = ngpu > 0 ? psa_init(local % ngpu, &ctx) : PSA_ERR_NO_DEVICE;
    if (st == PSA_OK) st = psa_set_seq1(ctx, seq1.c_str(), (int64_t)seq1.size());
    if (st == PSA_OK) st = psa_set_weights(ctx, weights);
    int worst = st;
    MPI_Allreduce(&st, &worst, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);  // nobody enters the all-gathers alone
    if (st != PSA_OK)
      std::fprintf(stderr, "psa_mpi: rank %d: %s (%s)\n", rank, psa_strerror(st), psa_last_error_text());
    int rc = worst == PSA_OK ? 0 : EXIT_FAILURE;
    MPI_Comm world = MPI_COMM_WORLD;
    size_t at = (size_t)header[0];
    for (int i = 0; i < n && rc == 0; ++i) {
      psa_result res;
      st = psa_score_range_allgather(ctx, blob.data() + at, len2[(size_t)i], rank, size, mpi_allgather, &world,
                                     stages, &res);
      if (st == PSA_OK && res.status == PSA_OK && stages == PSA_STAGE_ENUMERATE && res.n == 0 && res.candidates > 0) {
        // some shard's shortlist overflowed (every rank holds the same combined record and decides alike):
        // the resolve stage gives the float32 answer of this pair
        st = psa_score_range_allgather(ctx, blob.data() + at, len2[(size_t)i], rank, size, mpi_allgather, &world,
                                       PSA_STAGE_RESOLVE, &res);
      } else if (st == PSA_OK && res.status == PSA_OK && stages == (PSA_STAGE_ENUMERATE | PSA_STAGE_RESOLVE) &&
                 res.enum_check < 0) {
        st = PSA_ERR_CHECK;
      }
      at += (size_t)len2[(size_t)i];
      if (st != PSA_OK || res.status != PSA_OK) {  // the same on every rank: the records were all-gathered
        if (rank == 0)
          std::fprintf(stderr, "psa_mpi: Seq2 #%d: %s\n", i + 1, psa_strerror(st != PSA_OK ? st : res.status));
        rc = EXIT_FAILURE;
        break;
      }
      if (rank == 0) {
        char *line = psa_format_result(&res);
        std::printf("%s", line);
        std::fprintf(out, "%s", line);  // P/cFunctions.c:45
        std::free(line);
      }
    }
    psa_shutdown(ctx);
    if (rank == 0) {
      if (out) std::fclose(out);
      double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
      std::printf("Total Time: %1.4f\n", secs);  // P/main.c:97
    }
    MPI_Finalize();
    return rc;
  }

  // every rank computes the same split
  std::vector<int32_t> owner((size_t)n, 0);
  psa_partition_stages(n, len2.data(), (int64_t)seq1.size(), size, stages, owner.data());
  std::vector<std::string> mine_s;
  std::vector<int> mine_idx;
  {
    size_t at = (size_t)header[0];
    for (int i = 0; i < n; ++i) {
      if (owner[(size_t)i] == rank) {
        mine_s.emplace_back(blob.data() + at, (size_t)len2[(size_t)i]);
        mine_idx.push_back(i);
      }
      at += (size_t)len2[(size_t)i];
    }
  }
  std::vector<char>().swap(blob);
  if (!dry) std::printf("proc[%d] portion = %d\n", rank, (int)mine_s.size());  // P/main.c:64, :113

  // score my share on the GPU that goes with my place on this node
  std::vector<psa_result> mine_r(mine_s.size());
  int st = PSA_OK;
  if (!dry && !mine_s.empty()) {
    const int ngpu = psa_device_count();
    psa_ctx *ctx = nullptr;
    st = ngpu > 0 ? psa_init(local % ngpu, &ctx) : PSA_ERR_NO_DEVICE;
    if (st == PSA_OK) st = psa_set_seq1(ctx, seq1.c_str(), (int64_t)seq1.size());
    if (st == PSA_OK) st = psa_set_weights(ctx, weights);
    if (st == PSA_OK) {
      std::vector<const char *> ptr;
      std::vector<int32_t> lens;
      for (const auto &s : mine_s) {
        ptr.push_back(s.c_str());
        lens.push_back((int32_t)s.size());
      }
      st = psa_score_batch_answer(ctx, (int)ptr.size(), ptr.data(), lens.data(), stages, mine_r.data(), nullptr);
    }
    if (st != PSA_OK)
      std::fprintf(stderr, "psa_mpi: rank %d: %s (%s)\n", rank, psa_strerror(st), psa_last_error_text());
    psa_shutdown(ctx);
  }
  int worst = st;
  MPI_Allreduce(&st, &worst, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
  if (worst != PSA_OK) {  // the reference exits on device errors, P/cudaFunctions.cu:5-18
    if (out) std::fclose(out);
    MPI_Finalize();
    return EXIT_FAILURE;
  }

  // records back to rank 0, which knows who owns what
  std::vector<int> counts((size_t)size, 0), displs((size_t)size, 0);
  for (int i = 0; i < n; ++i) counts[(size_t)owner[(size_t)i]] += (int)sizeof(psa_result);
  for (int r = 1; r < size; ++r) displs[(size_t)r] = displs[(size_t)r - 1] + counts[(size_t)r - 1];
  std::vector<psa_result> all((size_t)n);
  MPI_Gatherv(mine_r.data(), (int)(mine_r.size() * sizeof(psa_result)), MPI_BYTE, all.data(), counts.data(),
              displs.data(), MPI_BYTE, 0, MPI_COMM_WORLD);
  int rc = 0;
  if (rank == 0) {
    std::vector<size_t> next((size_t)size);
    for (int r = 0; r < size; ++r) next[(size_t)r] = (size_t)displs[(size_t)r] / sizeof(psa_result);
    for (int i = 0; i < n && rc == 0; ++i) {
      const psa_result &res = all[next[(size_t)owner[(size_t)i]]++];
      if (dry) {
        std::printf("Seq2 #%d (%d characters) -> rank %d\n", i + 1, (int)len2[(size_t)i], (int)owner[(size_t)i]);
        continue;
      }
      if (res.status != PSA_OK) {
        std::fprintf(stderr, "psa_mpi: Seq2 #%d: %s\n", i + 1, psa_strerror(res.status));
        rc = EXIT_FAILURE;
        break;
      }
      char *line = psa_format_result(&res);
      std::printf("%s", line);
      std::fprintf(out, "%s", line);  // P/cFunctions.c:45
      std::free(line);
    }
    if (out) std::fclose(out);
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    std::printf("Total Time: %1.4f\n", secs);  // P/main.c:97
  }
  MPI_Bcast(&rc, 1, MPI_INT, 0, MPI_COMM_WORLD);
  MPI_Finalize();
  return rc;
}
