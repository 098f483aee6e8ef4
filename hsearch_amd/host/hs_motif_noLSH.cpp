// hs_motif_noLSH.cpp -- the `motif_both_points_noLSH` program of the reference (exhaustive search,
// the ground truth of the LSH search) on the GPU path.
//
// Keeps the reference's command line (hclust/src/hclust/motif_both_points_noLSH.cpp:95-107):
//     -d <db.points> -c <centers.points> -l <k> -T <R> -o <out>
// and exit behaviour.  The reference also writes every pair beyond R to <out>notlessthan.txt
// (:41-49; Q x N lines); here only with -notlessthan.  Addition: -G <GPU ordinal>.
#include <time.h>

#include <iostream>
#include <string>
#include <vector>

#include "hs_cli.hpp"
#include "hs_host.hpp"

int main(int argc, const char* argv[]) {
  const hs_cli::Opt opts[] = {
      {"db", 'd', "protein database file", true},
      {"center", 'c', "centers from Pfam database", true},
      {"len", 'l', "kmer length", true},
      {"threshold", 'T', "kmer threshold", true},
      {"output", 'o', "output file name", true},
      {"notlessthan", 'n', "also write the pairs beyond the threshold to <out>notlessthan.txt", false},
      {"device", 'G', "GPU ordinal [0]", false},
  };
  std::map<std::string, std::string> val;
  const int rc = hs_cli::Parse(argc, argv, opts, sizeof(opts) / sizeof(opts[0]), "cluster kmers to motifs",
                               nullptr, &val, "notlessthan");
  if (rc >= 0) return rc;
  const uint32_t len = (uint32_t)strtoul(val["len"].c_str(), nullptr, 10);
  const double hash_R = strtod(val["threshold"].c_str(), nullptr);
  const int device = val.count("device") ? atoi(val["device"].c_str()) : 0;
  try {
    std::vector<std::string> kmer_names, center_names;
    std::vector<hsearch::Point> kmers, centers;
    std::cout << "Read Kmers..." << std::endl;
    if (!hsearch::ReadPointsFile(val["db"], 8 * len, &kmer_names, &kmers)) {
      fprintf(stderr, "cannot open %s\n", val["db"].c_str());
      return EXIT_FAILURE;
    }
    std::cout << "Read Centers..." << std::endl;
    if (!hsearch::ReadPointsFile(val["center"], 8 * len, &center_names, &centers)) {
      fprintf(stderr, "cannot open %s\n", val["center"].c_str());
      return EXIT_FAILURE;
    }
    std::cout << "number of kmers " << kmers.size() << std::endl;
    std::cout << "number of centers " << centers.size() << std::endl;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    std::string err;
    const int st = hsearch::SearchBruteForce(kmers, centers, kmer_names, center_names, hash_R, val["output"],
                                             device, &err, val.count("notlessthan") != 0);
    if (st != 0) {
      fprintf(stderr, "ERROR: %s (status %d)\n", err.c_str(), st);
      return EXIT_FAILURE;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    printf("Searching takes %lf seconds\n", (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));
  } catch (const std::bad_alloc&) {
    fprintf(stderr, "ERROR: could not allocate memory\n");
    return EXIT_FAILURE;
  } catch (const std::exception& e) {
    fprintf(stderr, "%s\n", e.what());
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
