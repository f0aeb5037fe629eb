// Reading the reference's input file, shared by the host programs (psa_host, psa_mpi).
//
// Grammar of P/cFunctions.c:17-37 (fscanf with %f, %s, %d): whitespace-separated tokens -- four
// weights, Seq1, a count, then that many Seq2; whatever follows the count-th Seq2 is ignored (the
// reference's own input3.txt announces 6 and holds 7).  Buffers are dynamic: the reference caps Seq1
// at 5000 and Seq2 at 3000 characters (P/myProto.h:10-11).
#ifndef PSA_INPUT_H
#define PSA_INPUT_H

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace psa_host {

inline bool read_token(FILE *f, std::string &tok) {
  tok.clear();
  auto space = [](int c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; };
  int c;
  while ((c = std::fgetc(f)) != EOF && space(c)) {
  }
  if (c == EOF) return false;
  do tok.push_back((char)c);
  while ((c = std::fgetc(f)) != EOF && !space(c));
  return true;
}

struct Input {
  float weights[4] = {0, 0, 0, 0};
  std::string seq1;
  int count = 0;  // as announced by the file (what the reference prints, P/main.c:51)
  std::vector<std::string> seq2;
};

// false if the file cannot be opened (the caller prints the reference's message, P/cFunctions.c:21-24)
inline bool read_input(const char *name, Input &in) {
  FILE *f = std::fopen(name, "r");
  if (!f) return false;
  std::string tok;
  for (int i = 0; i < 4; ++i)
    if (read_token(f, tok)) in.weights[i] = std::strtof(tok.c_str(), nullptr);
  read_token(f, in.seq1);
  if (read_token(f, tok)) in.count = std::atoi(tok.c_str());
  in.seq2.resize((size_t)(in.count > 0 ? in.count : 0));
  for (auto &s : in.seq2) read_token(f, s);
  std::fclose(f);
  return true;
}

// "--stages=resolve|enumerate|both" or "--stages" followed by the word; returns the word or nullptr
inline const char *stages_option(int argc, char **argv, int &i) {
  if (std::strncmp(argv[i], "--stages=", 9) == 0) return argv[i] + 9;
  if (std::strcmp(argv[i], "--stages") == 0 && i + 1 < argc) return argv[++i];
  return nullptr;
}

}  // namespace psa_host

#endif
