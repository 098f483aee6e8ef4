// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
//
// Harness around the REAL reference Kernel-LSH class (acgtun/hsearch, pcluster/src/pcluster/lsh.hpp
// + lsh.cpp), compiled from the sources where they lie under /root/reference (oracle/Makefile ->
// oracle/_ref/libref_klsh.so).  The pcluster PROGRAM does not compile as shipped (SURVEY section 2
// row 12), its KLSH translation unit does; PreClustering's feature loop (pcluster.cpp:23-33) is
// driven from tools/gen_golden.py through ref_klsh_hash.  This file contains no reference code.
//   seam: KLSH's planes are private (lsh.hpp:40-47) -> `#define private public`.
#include <stdint.h>
#include <string.h>
#include <vector>

#define private public
#include "pcluster/src/pcluster/lsh.hpp"
#undef private

#define HS_REF_API extern "C" __attribute__((visibility("default")))

static KLSH* g_klsh = nullptr;
static uint32_t g_feat = 0, g_bits = 0;

// KLSH(feat, bits, sigma) exactly as PreClustering constructs it (pcluster.cpp:13-17: 8^3, 16, 0.2);
// the engine is default-seeded (lsh.hpp:49), so the planes are a pure function of libstdc++.
HS_REF_API void ref_klsh_create(uint32_t feat, uint32_t bits, double sigma) {
  delete g_klsh;
  g_klsh = new KLSH(feat, bits, sigma);
  g_feat = feat;
  g_bits = bits;
}
HS_REF_API void ref_klsh_planes(double* w, double* b, double* t) {
  for (uint32_t i = 0; i < g_bits; ++i) {
    memcpy(w + (size_t)i * g_feat, g_klsh->m_project_w[i].data(), sizeof(double) * g_feat);
    b[i] = g_klsh->m_project_b[i];
    t[i] = g_klsh->m_project_t[i];
  }
}
HS_REF_API uint64_t ref_klsh_hash(const double* feat) {
  std::vector<double> p(feat, feat + g_feat);
  return g_klsh->GetHashValue(p);
}
