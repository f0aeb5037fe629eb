// psa_host: the reference's host program (P/main.c) on top of libpsa_hip.
//
// Same files and formats: reads "input.txt" (four weights, Seq1, a count, that many
// Seq2; P/cFunctions.c:17-37), writes one line per Seq2 in input order to
// "output.txt" (P/cFunctions.c:157, P/main.c:79-94) and echoes the lines on stdout
// (P/cudaFunctions.cu:321).  What changes: buffers are dynamic (the reference caps
// Seq1 at 5000 and Seq2 at 3000 characters, P/myProto.h:10-11), and the two MPI ranks
// with a count-based split (P/main.c:62-76) become one worker thread per GPU of the
// node with a cost-based split.
//
//   psa_host [--stages=resolve|enumerate|both] [--sequential-format] [input.txt [output.txt [ngpus]]]
//
// --stages (default: PSA_STAGES in the environment, else resolve) chooses where the answer comes from:
// the float32 recurrence alone, the candidate loop that scores every (offset, mutant) candidate -- what
// every call of the reference runs, P/cudaFunctions.cu:290-296 -- with the float32 replay of the candidates
// near its maximum, or both with the two required to agree (psa_score_batch_answer).
// --sequential-format prints, instead of the Parallel program's line, the line the reference's Sequential
// Solution prints per Seq2 (S/main.c:82), and nothing else on stdout; output.txt is opened and left
// empty, as that program leaves it (S/main.c:28-32).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "psa.h"
#include "psa_input.h"

int main(int argc, char **argv) {
  bool seq_format = false;
  const char *stages_text = std::getenv("PSA_STAGES");
  std::vector<const char *> pos;
  for (int i = 1; i < argc; ++i) {
    if (std::strcmp(argv[i], "--sequential-format") == 0)
      seq_format = true;
    else if (const char *w = psa_host::stages_option(argc, argv, i))
      stages_text = w;
    else
      pos.push_back(argv[i]);
  }
  uint32_t stages = PSA_STAGE_RESOLVE;
  if (psa_parse_stages(stages_text, &stages) != PSA_OK) {
    std::fprintf(stderr, "psa_host: %s\n", psa_last_error_text());
    return 2;
  }
  const char *in_name = pos.size() > 0 ? pos[0] : "input.txt";    // P/myProto.h:8
  const char *out_name = pos.size() > 1 ? pos[1] : "output.txt";  // P/myProto.h:9
  auto t_start = std::chrono::steady_clock::now();

  if (!seq_format) std::printf("my rank is %d and i'm currently reading from the file\n", 0);  // P/main.c:37
  psa_host::Input input;
  if (!psa_host::read_input(in_name, input)) {
    std::printf("Error in opening the file\n");  // P/cFunctions.c:21-24
    return 1;
  }
  const float *weights = input.weights;
  const std::string &seq1 = input.seq1;
  const std::vector<std::string> &seq2 = input.seq2;
  const int count = input.count;

  FILE *out = std::fopen(out_name, "w");
  if (!out) {
    std::printf("Error in opening the file\n");  // P/main.c:46-49
    return 1;
  }
  if (!seq_format) std::printf("num of seq2 = %d\n", count);  // P/main.c:51

  int ngpu = psa_device_count();
  if (pos.size() > 2) ngpu = std::min(ngpu, std::atoi(pos[2]));
  if (ngpu < 1) {
    std::fprintf(stderr, "psa_host: no GPU visible (there is no CPU path)\n");
    return 1;
  }
  const int n = (int)seq2.size();
  std::vector<int32_t> len2((size_t)n), owner((size_t)n, 0);
  for (int i = 0; i < n; ++i) len2[(size_t)i] = (int32_t)seq2[(size_t)i].size();
  psa_partition_stages(n, len2.data(), (int64_t)seq1.size(), ngpu, stages, owner.data());

  if (!seq_format)
    for (int g = 0; g < ngpu; ++g)  // P/main.c:64, :113, one line per worker instead of per rank
      std::printf("proc[%d] portion = %d\n", g, (int)std::count(owner.begin(), owner.end(), g));

  std::vector<psa_result> results((size_t)n);
  std::vector<int> worker_status((size_t)ngpu, PSA_OK);
  std::vector<std::string> worker_error((size_t)ngpu);  // the library's error text is per thread
  std::vector<std::thread> workers;
  for (int g = 0; g < ngpu; ++g) {
    workers.emplace_back([&, g]() {
      std::vector<const char *> mine;
      std::vector<int32_t> lens;
      std::vector<int> idx;
      for (int i = 0; i < n; ++i)
        if (owner[(size_t)i] == g) {
          mine.push_back(seq2[(size_t)i].c_str());
          lens.push_back(len2[(size_t)i]);
          idx.push_back(i);
        }
      if (mine.empty()) return;
      psa_ctx *ctx = nullptr;
      int st = psa_init(g, &ctx);
      if (st == PSA_OK) st = psa_set_seq1(ctx, seq1.c_str(), (int64_t)seq1.size());
      if (st == PSA_OK) st = psa_set_weights(ctx, weights);
      std::vector<psa_result> r(mine.size());
      if (st == PSA_OK) st = psa_score_batch_answer(ctx, (int)mine.size(), mine.data(), lens.data(), stages, r.data(), nullptr);
      if (st == PSA_OK)
        for (size_t k = 0; k < idx.size(); ++k) results[(size_t)idx[k]] = r[k];
      worker_status[(size_t)g] = st;
      if (st != PSA_OK) worker_error[(size_t)g] = psa_last_error_text();
      psa_shutdown(ctx);
    });
  }
  for (auto &w : workers) w.join();
  for (int g = 0; g < ngpu; ++g)
    if (worker_status[(size_t)g] != PSA_OK) {
      std::fprintf(stderr, "psa_host: GPU %d: %s (%s)\n", g, psa_strerror(worker_status[(size_t)g]),
                   worker_error[(size_t)g].c_str());
      return EXIT_FAILURE;  // the reference exits on device errors, P/cudaFunctions.cu:5-18
    }
  for (int i = 0; i < n; ++i) {
    if (results[(size_t)i].status != PSA_OK) {
      std::fprintf(stderr, "psa_host: Seq2 #%d: %s\n", i + 1, psa_strerror(results[(size_t)i].status));
      return EXIT_FAILURE;
    }
    if (seq_format) {
      const psa_result &r = results[(size_t)i];
      std::printf("MS(%d,%d)\tbest score: %1.4f\tbest offset: %d\n", r.n, r.k, (double)r.score, r.offset);  // S/main.c:82
      continue;
    }
    char *line = psa_format_result(&results[(size_t)i]);
    if (!line) {
      std::fprintf(stderr, "psa_host: out of memory\n");
      return EXIT_FAILURE;
    }
    std::printf("%s", line);
    std::fprintf(out, "%s", line);  // P/cFunctions.c:45
    std::free(line);
  }
  std::fclose(out);
  double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  if (seq_format)
    std::printf("Hello World!\n");  // S/main.c:87
  else
    std::printf("Total Time: %1.4f\n", secs);  // P/main.c:97
  return 0;
}
