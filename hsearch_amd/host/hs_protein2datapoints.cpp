// hs_protein2datapoints.cpp -- the `protein2datapoints` program of the reference
// (hclust/src/hclust/protein2datapoints.cpp): FASTA database -> sampled k-mers as a points file,
// the `-d` input of motif_both_points / motif_both_points_noLSH.
//
// Keeps the reference's command line (:112-120): -d <proteins.fa> -l <k> -n <num of proteins out>
// -o <out>.  Additions: -s <seed of the window strides> [time], -Q 0|1: the reference's E <-> Q
// exchange [1: its ProteinDB stores AA20[base[c]] (protein.hpp:62); 0 embeds E as Glu].
// Host only: no GPU work (the search programs take the FASTA file directly; this tool exists for
// the reference's file-based workflow).
#include <time.h>

#include <iostream>
#include <string>

#include "hs_cli.hpp"
#include "hs_host.hpp"

int main(int argc, const char* argv[]) {
  const hs_cli::Opt opts[] = {
      {"db", 'd', "protein database file", true},
      {"len", 'l', "kmer length", true},
      {"nnn", 'n', "num of proteins out", true},
      {"output", 'o', "output file name", true},
      {"seed", 's', "seed of the window strides [time]", false},
      {"ref-compat-eq-swap", 'Q', "exchange E and Q like the reference's ProteinDB [1]", false},
  };
  std::map<std::string, std::string> val;
  const int rc = hs_cli::Parse(argc, argv, opts, sizeof(opts) / sizeof(opts[0]),
                               "protein sequences to data points", "HSEARCH v1.0", &val);
  if (rc >= 0) return rc;
  const uint32_t len = (uint32_t)strtoul(val["len"].c_str(), nullptr, 10);
  const uint32_t num = (uint32_t)strtoul(val["nnn"].c_str(), nullptr, 10);
  const uint32_t seed = val.count("seed") ? (uint32_t)strtoul(val["seed"].c_str(), nullptr, 10) : (uint32_t)time(NULL);
  const bool swap = !val.count("ref-compat-eq-swap") || atoi(val["ref-compat-eq-swap"].c_str()) != 0;
  try {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    hsearch::ProteinDB db;
    std::cout << "Read protein sequences from " << val["db"] << std::endl;  // protein.hpp:38
    if (!hsearch::ReadProteinFasta(val["db"], swap, &db)) {
      fprintf(stderr, "cannot open %s\n", val["db"].c_str());
      return EXIT_FAILURE;
    }
    std::cout << "number of proteins " << (db.start.empty() ? 0 : db.start.size() - 1) << std::endl;
    std::cout << "total length " << db.residues.size() << std::endl;
    std::cout << "protein to data points... " << std::endl;
    const int64_t n = hsearch::Protein2Datapoints(db, len, num, val["output"], seed);
    if (n < 0) {
      fprintf(stderr, "cannot write %s\n", val["output"].c_str());
      return EXIT_FAILURE;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    printf("It takes %lf seconds\n", (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));
  } catch (const std::bad_alloc&) {
    fprintf(stderr, "ERROR: could not allocate memory\n");
    return EXIT_FAILURE;
  } catch (const std::exception& e) {
    fprintf(stderr, "%s\n", e.what());
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
