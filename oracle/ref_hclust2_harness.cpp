// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
//
// Compiles the REAL reference greedy LSH clustering (acgtun/hsearch hclust/src/hclust/hclust2.cpp)
// from the sources under /root/reference and exposes it through a C ABI.  Same seams as
// ref_search_harness.cpp: private->public for LSH::a/b, std::random_device re-pointed at a counting
// seed source so the l-th table's LSH is seeded with (seed + l), main() renamed.  Contains no
// reference code, only the #include.  Built into oracle/_ref/libref_hclust2.so by oracle/Makefile.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cassert>
#include <cmath>
#include <fstream>
#include <iostream>
#include <iterator>
#include <limits>
#include <ostream>
#include <random>
#include <sstream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <dirent.h>
#include <errno.h>
#include <sys/stat.h>
#include <unistd.h>

namespace hs_ref_seam {
static unsigned int g_next_seed = 0;
struct CountingSeedSource {
  typedef unsigned int result_type;
  unsigned int operator()() { return g_next_seed++; }
};
}  // namespace hs_ref_seam

#define random_device hs_ref_seam::CountingSeedSource
#define private public
#define main hs_ref_unused_main
// -DHS_REF_HCLUST3 builds the same harness over hclust3.cpp, hclust2's twin that embeds a k-mer every
// time it is looked at instead of once (KMER::point(), hclust3.cpp:43-45): libref_hclust3.so
#ifdef HS_REF_HCLUST3
#include "hclust/src/hclust/hclust3.cpp"
#else
#include "hclust/src/hclust/hclust2.cpp"
#endif
#undef main
#undef private
#undef random_device

#define HS_REF_API extern "C" __attribute__((visibility("default")))

struct CoutMute {
  std::streambuf* old;
  std::ostringstream sink;
  CoutMute() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~CoutMute() { std::cout.rdbuf(old); }
};

// Planes as hclust2's `LSH lsh(DIMENSION, K, W)` (hclust2.cpp:104) draws them for table l after the
// seed counter was reset to `seed`: identical to ref_search_harness's ref_lsh_planes.
HS_REF_API void ref2_lsh_planes(uint32_t seed, uint32_t dim, uint32_t K, double W, uint32_t L,
                                double* a_out, double* b_out) {
  hs_ref_seam::g_next_seed = seed;
  for (uint32_t l = 0; l < L; ++l) {
    LSH lsh(dim, K, W);
    for (uint32_t k = 0; k < K; ++k) {
      memcpy(a_out + ((size_t)l * K + k) * dim, lsh.a[k].data(), sizeof(double) * dim);
      b_out[(size_t)l * K + k] = lsh.b[k];
    }
  }
}

// KmerToCoordinates (hclust2.cpp:49-62) on n sequences of klen letters each (seqs is n*klen chars,
// no separators).  out[n][8*klen].
HS_REF_API void ref2_kmer_to_coordinates(const char* seqs, uint64_t n, uint32_t klen, double* out) {
  DIMENSION = AACoordinateSize * klen;
  for (uint64_t i = 0; i < n; ++i) {
    Point p = KmerToCoordinates(std::string(seqs + i * klen, klen));
    memcpy(out + i * DIMENSION, p.data.data(), sizeof(double) * DIMENSION);
  }
}

// Clustering() (hclust2.cpp:86-151) on n k-mers named by their decimal index; writes the
// reference-format clusters file ("#clusterid:<i>:size<m>" + member names) to out_path.
HS_REF_API int ref2_clustering(uint32_t seed, const char* seqs, uint64_t n, uint32_t klen,
                               uint32_t K, uint32_t L, double W, double R, const char* out_path) {
  CoutMute mute;
  DIMENSION = AACoordinateSize * klen;
  std::vector<KMER> kmers;
  kmers.reserve(n);
  for (uint64_t i = 0; i < n; ++i) {
    std::string s(seqs + i * klen, klen);
#ifdef HS_REF_HCLUST3
    kmers.push_back(KMER(std::to_string(i), s));
#else
    kmers.push_back(KMER(std::to_string(i), s, KmerToCoordinates(s)));
#endif
  }
  hs_ref_seam::g_next_seed = seed;
  fflush(stdout);
  int saved = dup(1);
  FILE* devnull = fopen("/dev/null", "w");
  dup2(fileno(devnull), 1);  // Clustering() printf()s its timing
  Clustering(kmers, K, L, W, R, std::string(out_path));
  fflush(stdout);
  dup2(saved, 1);
  close(saved);
  fclose(devnull);
  return 0;
}
