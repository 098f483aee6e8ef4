// hs_center_distance_sampling.cpp -- the `centerDistanceSmapling` program of the reference
// (hclust/src/hclust/centerDistanceSmapling.cpp): motif families -> centroids, and the distance
// distributions around them.
//
// Keeps the reference's command line (:404-412): -k <families file> -d <protein points file>
// -l <k> -o <output prefix>.  As the reference's main() (:476-477) it runs
// sequencedatabase2centers: <dir>/<o>innercenter_protein_centers_0.txt and
// <dir>/<o>ramdom_protein_centers_0.txt (the centre x database distances on the GPU).  Additions:
// -format points writes the centroids as a points file <o>hclust.format.txt instead
// (cluster2datapoint, :110-136 -- the `-c` input of motif_both_points; no database needed, no GPU
// work), -m <min family size> [50], -D <dir> [./pro2centerdis], -G <GPU ordinal>.
#include <iostream>
#include <string>
#include <vector>

#include "hs_cli.hpp"
#include "hs_host.hpp"

int main(int argc, const char* argv[]) {
  const hs_cli::Opt opts[] = {
      {"kmers", 'k', "kmers file", true},
      {"protein", 'd', "protein file", false},
      {"len", 'l', "kmer length", true},
      {"output", 'o', "output file name", true},
      {"format", 'f', "distances (default) | points", false},
      {"minsize", 'm', "smallest family kept [50]", false},
      {"dir", 'D', "directory of the distance files [./pro2centerdis]", false},
      {"device", 'G', "GPU ordinal [0]", false},
  };
  std::map<std::string, std::string> val;
  const int rc = hs_cli::Parse(argc, argv, opts, sizeof(opts) / sizeof(opts[0]), "pairwiseDistanceSampling",
                               "pairwiseDistanceSampling v1.0", &val);
  if (rc >= 0) return rc;
  const uint32_t len = (uint32_t)strtoul(val["len"].c_str(), nullptr, 10);
  const uint32_t min_size = val.count("minsize") ? (uint32_t)strtoul(val["minsize"].c_str(), nullptr, 10) : 50;
  const int device = val.count("device") ? atoi(val["device"].c_str()) : 0;
  const std::string format = val.count("format") ? val["format"] : "distances";
  const std::string dir = val.count("dir") ? val["dir"] : "./pro2centerdis";
  try {
    std::vector<hsearch::MotifFamily> families;
    if (!hsearch::ReadMotifFamilies(val["kmers"], min_size, &families)) {
      fprintf(stderr, "cannot open %s\n", val["kmers"].c_str());
      return EXIT_FAILURE;
    }
    std::cout << "Number of Clusters: " << families.size() << std::endl;  // :458
    std::vector<hsearch::Point> centers;
    std::string err;
    if (!hsearch::FamilyCenters(families, len, &centers, &err)) {
      fprintf(stderr, "ERROR: %s\n", err.c_str());
      return EXIT_FAILURE;
    }
    if (format == "points") {
      if (!hsearch::Cluster2DataPoint(families, centers, val["output"])) {
        fprintf(stderr, "cannot write %shclust.format.txt\n", val["output"].c_str());
        return EXIT_FAILURE;
      }
      return EXIT_SUCCESS;
    }
    if (format != "distances" || !val.count("protein")) {
      fprintf(stderr, format != "distances" ? "unknown -format\n" : "missing required option -d\n");
      return format != "distances" ? EXIT_FAILURE : EXIT_SUCCESS;
    }
    std::vector<std::string> names;
    std::vector<hsearch::Point> kmers_proteins;
    if (!hsearch::ReadPointsFile(val["protein"], 8 * len, &names, &kmers_proteins)) {
      fprintf(stderr, "cannot open %s\n", val["protein"].c_str());
      return EXIT_FAILURE;
    }
    const int st = hsearch::SequenceDatabase2Centers(kmers_proteins, centers, val["output"], dir, device, &err);
    if (st != 0) {
      fprintf(stderr, "ERROR: %s (status %d)\n", err.c_str(), st);
      return EXIT_FAILURE;
    }
  } catch (const std::bad_alloc&) {
    fprintf(stderr, "ERROR: could not allocate memory\n");
    return EXIT_FAILURE;
  } catch (const std::exception& e) {
    fprintf(stderr, "%s\n", e.what());
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
