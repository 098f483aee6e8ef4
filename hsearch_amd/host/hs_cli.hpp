// hs_cli.hpp -- the command-line convention of the reference's programs (smithlab OptionParser as
// they use it): single-dash short and long names, every option takes a value, `-help`/`-?` or no
// argument at all prints the help and exits 0, and so does a missing required option
// (e.g. motif_both_points_noLSH.cpp:108-119).
#ifndef HS_CLI_HPP
#define HS_CLI_HPP

#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <string>

namespace hs_cli {

struct Opt {
  const char* long_name;
  char short_name;
  const char* descr;
  bool required;
};

inline void Help(const char* prog, const Opt* opts, size_t n, const char* about) {
  fprintf(stderr, "Usage: %s [OPTIONS]\n\nOptions:\n", prog);
  for (size_t i = 0; i < n; ++i)
    fprintf(stderr, "  -%c, -%-12s %s%s\n", opts[i].short_name, opts[i].long_name, opts[i].descr,
            opts[i].required ? " [REQUIRED]" : "");
  fprintf(stderr, "\nHelp options:\n  -?, -help   print this help message\n\n%s\n", about);
}

// Returns -1 to go on, otherwise the exit code.  Flags (options without a value) are the names in
// `flags`, a space-separated list.
inline int Parse(int argc, const char* argv[], const Opt* opts, size_t n, const char* about,
                 const char* banner, std::map<std::string, std::string>* val,
                 const std::string& flags = "") {
  bool help = false;
  for (int i = 1; i < argc; ++i) {
    std::string arg = argv[i];
    if (arg == "-help" || arg == "--help" || arg == "-?" || arg == "-about") {
      help = true;
      continue;
    }
    if (arg.size() < 2 || arg[0] != '-') continue;
    const std::string name = arg.substr(arg[1] == '-' ? 2 : 1);
    const Opt* hit = nullptr;
    for (size_t o = 0; o < n; ++o)
      if (name == opts[o].long_name || (name.size() == 1 && name[0] == opts[o].short_name)) hit = &opts[o];
    if (!hit) {
      fprintf(stderr, "unknown option %s\n", arg.c_str());
      return EXIT_FAILURE;
    }
    if ((" " + flags + " ").find(" " + std::string(hit->long_name) + " ") != std::string::npos) {
      (*val)[hit->long_name] = "1";
      continue;
    }
    if (i + 1 >= argc) {
      fprintf(stderr, "option %s needs a value\n", arg.c_str());
      return EXIT_FAILURE;
    }
    (*val)[hit->long_name] = argv[++i];
  }
  if (argc > 1 && !help && banner) {
    fprintf(stdout, "[WELCOME TO %s -- MI355X]\n[%s", banner, argv[0]);
    for (int i = 1; i < argc; ++i) fprintf(stdout, " %s", argv[i]);
    fprintf(stdout, "]\n");
  }
  if (argc == 1 || help) {
    Help(argv[0], opts, n, about);
    return EXIT_SUCCESS;
  }
  for (size_t o = 0; o < n; ++o)
    if (opts[o].required && !val->count(opts[o].long_name)) {
      fprintf(stderr, "missing required option -%c\n", opts[o].short_name);
      Help(argv[0], opts, n, about);
      return EXIT_SUCCESS;
    }
  return -1;
}

}  // namespace hs_cli

#endif
