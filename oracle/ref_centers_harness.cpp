// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
//
// Compiles the REAL reference centroid builder (acgtun/hsearch
// hclust/src/hclust/centerDistanceSmapling.cpp) from the sources under /root/reference.  Contains
// no reference code, only the #include.  Built into oracle/_ref/libref_centers.so by oracle/Makefile.
#include "ref_tools_common.h"

#define main hs_ref_centers_main
#include "hclust/src/hclust/centerDistanceSmapling.cpp"
#undef main

// The program as it runs: `centerDistanceSmapling -k <families> -d <points> -l <k> -o <out>` with
// the working directory set to `workdir` (which must hold a directory pro2centerdis/, :149,174).
// The points file must hold at least 100000 points (:166-173 reads that many unconditionally).
HS_REF_API int refc_main(const char* workdir, const char* families_path, const char* points_path,
                         uint32_t kmer_length, const char* out_name) {
  HsRefCoutMute mute;
  char old[4096];
  if (!getcwd(old, sizeof(old)) || chdir(workdir) != 0) return -1;
  const std::string len = std::to_string(kmer_length);
  const char* argv[] = {"centerDistanceSmapling", "-k", families_path, "-d", points_path,
                        "-l", len.c_str(), "-o", out_name, nullptr};
  FILE* saved = stdout;
  stdout = fopen("/dev/null", "w");  // the banner goes through fprintf(stdout, ...)
  const int rc = hs_ref_centers_main(9, argv);
  fclose(stdout);
  stdout = saved;
  if (chdir(old) != 0) return -2;
  return rc;
}

// cluster2datapoint() (:110-136) over families given as arrays: names[f], members of family f are
// seqs[first[f] .. first[f + 1]).  Writes <out_prefix>hclust.format.txt.
HS_REF_API int refc_cluster2datapoint(uint32_t kmer_length, uint32_t n_families, const char* const* names,
                                      const uint32_t* first, const char* const* seqs,
                                      const char* out_prefix) {
  HsRefCoutMute mute;
  DIMENSION = AACoordinateSize * kmer_length;
  std::vector<std::pair<std::string, std::vector<std::string> > > clusters(n_families);
  for (uint32_t f = 0; f < n_families; ++f) {
    clusters[f].first = names[f];
    for (uint32_t m = first[f]; m < first[f + 1]; ++m) clusters[f].second.push_back(seqs[m]);
  }
  cluster2datapoint(clusters, std::string(out_prefix));
  return 0;
}
