// hs_pcluster_pregroup.cpp -- the Kernel-LSH front half of the reference's `pcluster` program
// (pcluster/src/pcluster/pcluster.cpp:11-81, PreClustering) on the GPU path.  The alignment back
// half of pcluster is out of scope.
//     -d <proteins.fa>   protein database (FASTA, pcluster.cpp:122-124)
//     -o <out>           pre-groups: one line per protein, "<code>\t<name>", groups in ascending
//                        code order, proteins in file order inside a group
//     --device <n>       GPU ordinal [0];  --unknown-seed <s>  seed for letters outside the alphabet
// Prints "[NUMBER OF PRE-GROUPS n]" to stderr like the reference (pcluster.cpp:36).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "hs_host.hpp"

int main(int argc, const char* argv[]) {
  std::string db_path, out_path;
  int device = 0;
  uint32_t unknown_seed = 0;
  for (int i = 1; i < argc; ++i) {
    const char* a = argv[i];
    while (*a == '-') ++a;
    const bool has = i + 1 < argc;
    if ((!strcmp(a, "d") || !strcmp(a, "database")) && has) db_path = argv[++i];
    else if ((!strcmp(a, "o") || !strcmp(a, "output")) && has) out_path = argv[++i];
    else if (!strcmp(a, "device") && has) device = atoi(argv[++i]);
    else if (!strcmp(a, "unknown-seed") && has) unknown_seed = (uint32_t)strtoul(argv[++i], nullptr, 10);
    else if (!strcmp(a, "help") || !strcmp(a, "?")) db_path.clear(), out_path.clear();
  }
  if (db_path.empty() || out_path.empty()) {
    fprintf(stderr, "Usage: %s -d <proteins.fa> -o <out> [--device n] [--unknown-seed s]\n", argv[0]);
    return EXIT_SUCCESS;  // the reference prints its help and exits 0 on a missing option
  }
  hsearch::PclusterDB db;
  if (!hsearch::ReadPclusterFasta(db_path, unknown_seed, &db)) {
    fprintf(stderr, "cannot open input file %s\n", db_path.c_str());
    return EXIT_FAILURE;
  }
  std::map<uint64_t, std::vector<uint32_t> > buckets;
  std::string err;
  const int st = hsearch::PreClustering(db, device, &buckets, &err);
  if (st != 0) {
    fprintf(stderr, "ERROR: %s (status %d)\n", err.c_str(), st);
    return EXIT_FAILURE;
  }
  fprintf(stderr, "[NUMBER OF PRE-GROUPS %lu]\n", (unsigned long)buckets.size());
  std::ofstream fout(out_path.c_str());
  for (const auto& kv : buckets)
    for (uint32_t i : kv.second)
      fout << kv.first << "\t" << (i < db.names.size() ? db.names[i] : std::string()) << "\n";
  return EXIT_SUCCESS;
}
