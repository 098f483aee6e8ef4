// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
//
// Harness that compiles the REAL reference search path (acgtun/hsearch, hclust/src/hclust) from the
// sources where they lie under /root/reference and exposes it through a small C ABI, so that
//   (a) the CPU restatement in oracle/hs_oracle.cpp can be validated against it,
//   (b) golden vectors under tests/golden/ can be generated from it (tools/gen_golden.py),
//   (c) bench.py can time it as cpu_baseline.kind == "reference".
// This file contains no reference code: it only #includes the reference translation unit.  It is
// built by oracle/Makefile into oracle/_ref/libref_search.so (git-ignored, travels with gpurun).
//
// Two seams are needed to drive the reference deterministically:
//   * LSH::a / LSH::b are private (lsh.hpp:65-66)            -> `#define private public`
//   * LSH::LSH seeds from std::random_device (lsh.hpp:19-20) -> `random_device` is re-pointed at a
//     counter-based stand-in, so the l-th LSH object constructed after ref_seed_reset(s) is seeded
//     with (s + l).  The planes are still drawn by the reference's own constructor with
//     libstdc++'s default_random_engine / normal_distribution / uniform_real_distribution.
//   * main() of motif_both_points.cpp is renamed; Search() (motif_both_points.cpp:195-250) is
//     called as-is.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <fstream>
#include <iostream>
#include <random>
#include <sstream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

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
#include "hclust/src/hclust/motif_both_points.cpp"
#undef main
#undef private
#undef random_device

#define HS_REF_API extern "C" __attribute__((visibility("default")))

// Silence the reference's progress chatter ("table size ...") while a call runs.
struct CoutMute {
  std::streambuf* old;
  std::ostringstream sink;
  CoutMute() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~CoutMute() { std::cout.rdbuf(old); }
};

HS_REF_API void ref_seed_reset(uint32_t seed) { hs_ref_seam::g_next_seed = seed; }

// Planes exactly as `L` consecutive `LSH(dim, K, W)` constructions draw them after
// ref_seed_reset(seed).  a_out[L][K][dim], b_out[L][K].
HS_REF_API void ref_lsh_planes(uint32_t seed, uint32_t dim, uint32_t K, double W, uint32_t L,
                               double* a_out, double* b_out) {
  ref_seed_reset(seed);
  for (uint32_t l = 0; l < L; ++l) {
    LSH lsh(dim, K, W);
    for (uint32_t k = 0; k < K; ++k) {
      memcpy(a_out + ((size_t)l * K + k) * dim, lsh.a[k].data(), sizeof(double) * dim);
      b_out[(size_t)l * K + k] = lsh.b[k];
    }
  }
}

// Reference DotProduct / HashBucketIndex / HashKey (lsh.hpp:33-59) with INJECTED planes of one
// table.  pts[n][dim]; dots_out[n][K] (may be NULL); buckets_out[n][K]; keys_out: n strings of
// key_stride bytes each, NUL-terminated (may be NULL).
HS_REF_API void ref_hash(const double* a, const double* b, uint32_t dim, uint32_t K, double W,
                         const double* pts, uint64_t n, double* dots_out, int32_t* buckets_out,
                         char* keys_out, uint32_t key_stride) {
  ref_seed_reset(0);
  LSH lsh(dim, K, W);
  for (uint32_t k = 0; k < K; ++k) {
    lsh.a[k].assign(a + (size_t)k * dim, a + (size_t)(k + 1) * dim);
    lsh.b[k] = b[k];
  }
  std::vector<double> p(dim);
  for (uint64_t i = 0; i < n; ++i) {
    p.assign(pts + i * dim, pts + (i + 1) * dim);
    for (uint32_t k = 0; k < K; ++k) {
      if (dots_out) dots_out[i * K + k] = lsh.DotProduct(p, k);
      buckets_out[i * K + k] = lsh.HashBucketIndex(p, k);
    }
    if (keys_out) {
      std::string key = lsh.HashKey(p);
      strncpy(keys_out + i * key_stride, key.c_str(), key_stride - 1);
      keys_out[i * key_stride + key_stride - 1] = 0;
    }
  }
}

// The reference Search() (motif_both_points.cpp:195-250), planes drawn by its own LSH constructor
// from seeds seed, seed+1, ... (== ref_lsh_planes(seed, ...)).  Names are the decimal indices, so
// the hits file reads "q id dist".
HS_REF_API int ref_search(uint32_t seed, uint32_t dim, const double* db, uint64_t n,
                          const double* centers, uint64_t q, uint32_t K, uint32_t L, double W,
                          double R, const char* out_path) {
  CoutMute mute;
  DIMENSION = dim;
  KMERLENGTH = dim / AACoordinateSize;
  std::vector<Point> kmers(n), cents(q);
  std::vector<std::string> kn(n), cn(q);
  for (uint64_t i = 0; i < n; ++i) {
    kmers[i].data.assign(db + i * dim, db + (i + 1) * dim);
    kn[i] = std::to_string(i);
  }
  for (uint64_t i = 0; i < q; ++i) {
    cents[i].data.assign(centers + i * dim, centers + (i + 1) * dim);
    cn[i] = std::to_string(i);
  }
  ref_seed_reset(seed);
  Search(kmers, cents, kn, cn, K, L, W, R, std::string(out_path));
  return 0;
}

// Search() once more, with the time at which its build loop ended.  The reference announces every finished
// table on std::cout and ends the line with std::endl (motif_both_points.cpp:217), i.e. with a flush: a stream
// buffer that notes the clock at every flush sees the moment the L-th table is done without touching the
// function.  *t_build_s = entry of Search() .. that moment (the build loop, :206-218), *t_rest_s = that moment
// .. return of Search() (the query loop :224-245, plus the destruction of the tables, which a call with zero
// centres measures alone: bench.py subtracts it).
#include <chrono>
struct FlushClock : std::streambuf {
  std::chrono::steady_clock::time_point last;
  unsigned flushes = 0;
  int overflow(int c) override { return c; }
  int sync() override {
    last = std::chrono::steady_clock::now();
    ++flushes;
    return 0;
  }
};
HS_REF_API int ref_search_timed(uint32_t seed, uint32_t dim, const double* db, uint64_t n,
                                const double* centers, uint64_t q, uint32_t K, uint32_t L, double W,
                                double R, const char* out_path, double* t_build_s, double* t_rest_s) {
  DIMENSION = dim;
  KMERLENGTH = dim / AACoordinateSize;
  std::vector<Point> kmers(n), cents(q);
  std::vector<std::string> kn(n), cn(q);
  for (uint64_t i = 0; i < n; ++i) {
    kmers[i].data.assign(db + i * dim, db + (i + 1) * dim);
    kn[i] = std::to_string(i);
  }
  for (uint64_t i = 0; i < q; ++i) {
    cents[i].data.assign(centers + i * dim, centers + (i + 1) * dim);
    cn[i] = std::to_string(i);
  }
  ref_seed_reset(seed);
  FlushClock fc;
  std::streambuf* old = std::cout.rdbuf(&fc);
  const auto t0 = std::chrono::steady_clock::now();
  Search(kmers, cents, kn, cn, K, L, W, R, std::string(out_path));
  const auto t1 = std::chrono::steady_clock::now();
  std::cout.rdbuf(old);
  if (fc.flushes != L) return 1;  // not the print pattern this timing relies on
  *t_build_s = std::chrono::duration<double>(fc.last - t0).count();
  *t_rest_s = std::chrono::duration<double>(t1 - fc.last).count();
  return 0;
}

// Same, but split into build and query so bench.py can time the two phases of the reference
// separately.  The reference has no such split (tables are locals of Search()); this re-runs the
// reference's own statements for each phase through its public pieces: LSH::HashKey for the build
// loop (motif_both_points.cpp:212-218) is exercised via Search() on zero centers.
HS_REF_API int ref_search_build_only(uint32_t seed, uint32_t dim, const double* db, uint64_t n,
                                     uint32_t K, uint32_t L, double W, const char* out_path) {
  return ref_search(seed, dim, db, n, db, 0, K, L, W, 1.0, out_path);
}

// Reference brute-force distance (motif_both_points.cpp:167-183): dist2_out[q][n] squared.
HS_REF_API void ref_pairwise_square(uint32_t dim, const double* db, uint64_t n,
                                    const double* centers, uint64_t q, double* dist2_out) {
  DIMENSION = dim;
  Point a, c;
  for (uint64_t i = 0; i < q; ++i) {
    c.data.assign(centers + i * dim, centers + (i + 1) * dim);
    for (uint64_t j = 0; j < n; ++j) {
      a.data.assign(db + j * dim, db + (j + 1) * dim);
      dist2_out[i * n + j] = PairwiseDistance_square(a, c);
    }
  }
}

// Reference evaulate() (motif_both_points.cpp:100-165): weighted recall of a hits file against a
// sorted ground-truth file.
HS_REF_API double ref_evaluate(const char* ground_truth, const char* hits, double R) {
  CoutMute mute;
  return evaulate(std::string(ground_truth), std::string(hits), R);
}

// The embedding constants (util.hpp:21-64,92) as the reference holds them.
HS_REF_API void ref_constants(double* coords_out /*[20][8]*/, double* dist2_out /*[20][20]*/,
                              int32_t* base_out /*[26]*/) {
  memcpy(coords_out, coordinates, sizeof(coordinates));
  memcpy(dist2_out, DISTANCE_SQUARE, sizeof(DISTANCE_SQUARE));
  for (int i = 0; i < 26; ++i) base_out[i] = base[i];
}
