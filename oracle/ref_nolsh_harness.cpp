// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
//
// Compiles the REAL reference exhaustive search (acgtun/hsearch
// hclust/src/hclust/motif_both_points_noLSH.cpp) from the sources under /root/reference and
// exposes its Search() through a C ABI.  Contains no reference code, only the #include.  Built
// into oracle/_ref/libref_nolsh.so by oracle/Makefile.
#include "ref_tools_common.h"

#define main hs_ref_unused_main
#include "hclust/src/hclust/motif_both_points_noLSH.cpp"
#undef main

// Search() of motif_both_points_noLSH.cpp:36-56 over db[n][dim], centers[q][dim]; names are
// "k<i>" and "c<i>".  Writes out_path and out_path + "notlessthan.txt" as the reference does.
HS_REF_API int refn_search(uint32_t dim, const double* db, uint64_t n, const double* centers,
                           uint64_t q, double R, const char* out_path) {
  HsRefCoutMute mute;
  DIMENSION = dim;
  KMERLENGTH = dim / AACoordinateSize;
  std::vector<Point> kmers(n), cents(q);
  std::vector<std::string> kn(n), cn(q);
  for (uint64_t i = 0; i < n; ++i) {
    kmers[i].data.assign(db + i * dim, db + (i + 1) * dim);
    kn[i] = "k" + std::to_string(i);
  }
  for (uint64_t i = 0; i < q; ++i) {
    cents[i].data.assign(centers + i * dim, centers + (i + 1) * dim);
    cn[i] = "c" + std::to_string(i);
  }
  Search(kmers, cents, kn, cn, R, std::string(out_path));
  return 0;
}
