// hs_hclust2.cpp -- the `hclust2` program of the reference on the GPU path.
//
// Keeps the reference's command line (hclust/src/hclust/hclust2.cpp:199-213):
//     -k <kmers.fa> -l <k> -K <hash_K> -L <hash_L> -W <w> -T <R> -o <out>
// (also -kmers/-len/-hash_K/-hash_L/-window/-threshold/-output) and its exit behaviour (missing
// option -> help, exit 0, :223-226; runtime error -> stderr, exit 1).  Additions: --seed (planes
// drawn like the reference's LSH constructor, table l seeded seed + l; default random_device as
// the reference) and --device.
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include <iostream>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "hs_host.hpp"

namespace {
struct Opt {
  const char* long_name;
  char short_name;
  const char* descr;
  bool required;
};
const Opt kOpts[] = {
    {"kmers", 'k', "kmers file", true},
    {"len", 'l', "kmer length", true},
    {"hash_K", 'K', "number of random lines", true},
    {"hash_L", 'L', "number of hash tables", true},
    {"window", 'W', "bucket width", true},
    {"threshold", 'T', "clustering threshold", true},
    {"output", 'o', "output file name", true},
    {"seed", 's', "seed of the LSH planes [random_device]", false},
    {"device", 'G', "GPU ordinal [0]", false},
};
void Help(const char* prog) {
  fprintf(stderr, "Usage: %s [OPTIONS]\n\nOptions:\n", prog);
  for (const Opt& o : kOpts)
    fprintf(stderr, "  -%c, -%-12s %s%s\n", o.short_name, o.long_name, o.descr,
            o.required ? " [REQUIRED]" : "");
  fprintf(stderr, "\nHelp options:\n  -?, -help   print this help message\n\ncluster kmers to motifs\n");
}
}  // namespace

int main(int argc, const char* argv[]) {
  bool help = false;
  std::map<std::string, std::string> val;
  for (int i = 1; i < argc; ++i) {
    std::string arg = argv[i];
    if (arg == "-help" || arg == "--help" || arg == "-?" || arg == "-about") {
      help = true;
      continue;
    }
    if (arg.size() < 2 || arg[0] != '-') continue;
    std::string name = arg.substr(arg[1] == '-' ? 2 : 1);
    const Opt* hit = nullptr;
    for (const Opt& o : kOpts)
      if (name == o.long_name || (name.size() == 1 && name[0] == o.short_name)) hit = &o;
    if (!hit) {
      fprintf(stderr, "unknown option %s\n", arg.c_str());
      return EXIT_FAILURE;
    }
    if (i + 1 >= argc) {
      fprintf(stderr, "option %s needs a value\n", arg.c_str());
      return EXIT_FAILURE;
    }
    val[hit->long_name] = argv[++i];
  }
  if (argc > 1 && !help) {
    fprintf(stdout, "[WELCOME TO PMF v1.0 -- MI355X]\n[%s", argv[0]);
    for (int i = 1; i < argc; ++i) fprintf(stdout, " %s", argv[i]);
    fprintf(stdout, "]\n");
  }
  if (argc == 1 || help) {
    Help(argv[0]);
    return EXIT_SUCCESS;
  }
  for (const Opt& o : kOpts)
    if (o.required && !val.count(o.long_name)) {
      fprintf(stderr, "missing required option -%c\n", o.short_name);
      Help(argv[0]);
      return EXIT_SUCCESS;
    }
  const uint32_t len = (uint32_t)strtoul(val["len"].c_str(), nullptr, 10);
  const uint32_t hash_K = (uint32_t)strtoul(val["hash_K"].c_str(), nullptr, 10);
  const uint32_t hash_L = (uint32_t)strtoul(val["hash_L"].c_str(), nullptr, 10);
  const double hash_W = strtod(val["window"].c_str(), nullptr);
  const double hash_R = strtod(val["threshold"].c_str(), nullptr);
  const int device = val.count("device") ? atoi(val["device"].c_str()) : 0;
  uint32_t seed;
  if (val.count("seed")) {
    seed = (uint32_t)strtoul(val["seed"].c_str(), nullptr, 10);
  } else {
    std::random_device rd;
    seed = rd();
  }
  try {
    std::vector<hsearch::Kmer> kmers;
    if (!hsearch::ReadKmerFasta(val["kmers"], &kmers)) {
      fprintf(stderr, "cannot open %s\n", val["kmers"].c_str());
      return EXIT_FAILURE;
    }
    printf("The number of kmers is %zu\n", kmers.size());
    const hsearch::Planes planes = hsearch::DrawPlanes(8 * len, hash_K, hash_L, hash_W, seed);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    std::cout << "Clustering... " << std::endl;
    std::string err;
    uint64_t n_clusters = 0;
    const int st = hsearch::Clustering(kmers, hash_K, hash_L, hash_W, hash_R, val["output"], planes,
                                       device, seed, &err, &n_clusters);
    if (st != 0) {
      fprintf(stderr, "ERROR: %s (status %d)\n", err.c_str(), st);
      return EXIT_FAILURE;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    printf("num_of_clusters = %llu\n", (unsigned long long)n_clusters);
    printf("Clustering takes %lf seconds\n", (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));
  } catch (const std::bad_alloc&) {
    fprintf(stderr, "ERROR: could not allocate memory\n");
    return EXIT_FAILURE;
  } catch (const std::exception& e) {
    fprintf(stderr, "%s\n", e.what());
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
