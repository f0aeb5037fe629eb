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
//   psa_host [input.txt [output.txt [ngpus]]]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "psa.h"

static bool read_token(FILE *f, std::string &tok) {
  tok.clear();
  int c;
  while ((c = std::fgetc(f)) != EOF && (c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f')) {
  }
  if (c == EOF) return false;
  do tok.push_back((char)c);
  while ((c = std::fgetc(f)) != EOF && !(c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'));
  return true;
}

int main(int argc, char **argv) {
  const char *in_name = argc > 1 ? argv[1] : "input.txt";    // P/myProto.h:8
  const char *out_name = argc > 2 ? argv[2] : "output.txt";  // P/myProto.h:9
  auto t_start = std::chrono::steady_clock::now();

  FILE *in = std::fopen(in_name, "r");
  if (!in) {
    std::printf("Error in opening the file\n");  // P/cFunctions.c:21-24
    return 1;
  }
  float weights[4] = {0, 0, 0, 0};
  std::string tok, seq1;
  for (int i = 0; i < 4; ++i)
    if (read_token(in, tok)) weights[i] = std::strtof(tok.c_str(), nullptr);
  read_token(in, seq1);
  int count = 0;
  if (read_token(in, tok)) count = std::atoi(tok.c_str());
  std::vector<std::string> seq2((size_t)(count > 0 ? count : 0));
  for (auto &s : seq2) read_token(in, s);  // tokens after the count-th Seq2 are ignored, as in the reference
  std::fclose(in);

  FILE *out = std::fopen(out_name, "w");
  if (!out) {
    std::printf("Error in opening the file\n");  // P/main.c:46-49
    return 1;
  }
  std::printf("num of seq2 = %d\n", count);  // P/main.c:51

  int ngpu = psa_device_count();
  if (argc > 3) ngpu = std::min(ngpu, std::atoi(argv[3]));
  if (ngpu < 1) {
    std::fprintf(stderr, "psa_host: no GPU visible (there is no CPU path)\n");
    return 1;
  }
  const int n = (int)seq2.size();
  std::vector<int32_t> len2((size_t)n), owner((size_t)n, 0);
  for (int i = 0; i < n; ++i) len2[(size_t)i] = (int32_t)seq2[(size_t)i].size();
  psa_partition_stages(n, len2.data(), (int64_t)seq1.size(), ngpu, PSA_STAGE_RESOLVE, owner.data());

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
      if (st == PSA_OK) st = psa_score_batch(ctx, (int)mine.size(), mine.data(), lens.data(), PSA_STAGE_RESOLVE, r.data());
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
  std::printf("Total Time: %1.4f\n", secs);  // P/main.c:97
  return 0;
}
