// hs_cluster.hip -- Clustering() of the reference (hclust/src/hclust/hclust2.cpp:86-151) on top of
// the C ABI: per table an index over the k-mers that are not absorbed yet and one GPU self-join
// (hs_self_join: all within-bucket pairs with sqrt(d2) <= R, exact fp64), then the greedy leader
// pass on the host.
//
// Why the host pass is exact and order-independent across buckets: buckets of one table are
// disjoint and the reference's inner loops (hclust2.cpp:107-132) read and write merged[] /
// clusters[] only for members of the bucket being processed, so the std::unordered_map iteration
// order cannot change the result; only the ascending-id order INSIDE a bucket matters.  Edges only
// join bucket mates, hence one global ascending pass over the active k-mers is the same
// computation as the per-bucket passes.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/hsearch.h"

namespace {

void set_err(char* err, uint32_t cap, const std::string& msg) {
  if (err && cap) {
    strncpy(err, msg.c_str(), cap - 1);
    err[cap - 1] = 0;
  }
}

}  // namespace

extern "C" hs_status hs_clustering(const hs_params* params, const double* a, const double* b,
                                   const double* coords, const uint8_t* codes, uint64_t n, double R,
                                   uint8_t* merged, uint32_t* owner, uint32_t* absorbed_table,
                                   char* err, uint32_t err_cap) {
  if (!params || !a || !b || (n && (!codes || !merged || !owner || !absorbed_table))) return HS_ERR_INVALID;
  const uint32_t k = params->k, K = params->K, L = params->L, d = 8 * k;
  for (uint64_t i = 0; i < n; ++i) {
    merged[i] = 0;  // hclust2.cpp:92-96
    owner[i] = (uint32_t)i;
    absorbed_table[i] = 0xffffffffu;
  }
  std::vector<uint32_t> active;
  std::vector<uint8_t> act_codes;
  std::vector<uint8_t> pre(n), pushed(n);
  std::vector<uint32_t> ei, ej;
  std::vector<double> ed;
  for (uint32_t l = 0; l < L; ++l) {
    // BuildLSHTalbe (hclust2.cpp:74-84): every k-mer with merged != 2, ascending id
    active.clear();
    for (uint64_t i = 0; i < n; ++i)
      if (merged[i] != 2) active.push_back((uint32_t)i);
    act_codes.resize(active.size() * (size_t)k);
    for (size_t t = 0; t < active.size(); ++t)
      memcpy(&act_codes[t * k], codes + (size_t)active[t] * k, k);
    hs_params p1 = *params;
    p1.L = 1;  // a fresh family per table (hclust2.cpp:104): planes of table l only
    hs_handle* h = nullptr;
    hs_status st = hs_create(&p1, a + (size_t)l * K * d, b + (size_t)l * K, coords, &h);
    if (st != HS_OK) {
      set_err(err, err_cap, std::string("hs_create: ") + (h ? hs_last_error(h) : "no usable gfx950 device"));
      hs_destroy(h);
      return st;
    }
    st = hs_index_build(h, act_codes.data(), active.size());
    uint64_t n_edges = 0;
    if (st == HS_OK) {
      uint64_t cap = std::max<uint64_t>(ei.size(), 4 * active.size() + 1024);
      for (;;) {
        ei.resize(cap);
        ej.resize(cap);
        ed.resize(cap);
        st = hs_self_join(h, R, /*sqrt_test=*/1, ei.data(), ej.data(), nullptr, ed.data(), cap, &n_edges);
        if (st == HS_ERR_CAPACITY) {
          cap = n_edges;
          continue;
        }
        break;
      }
    }
    if (st != HS_OK) {
      set_err(err, err_cap, std::string("table ") + std::to_string(l) + ": " + hs_last_error(h));
      hs_destroy(h);
      return st;
    }
    hs_destroy(h);
    // greedy pass (hclust2.cpp:107-132).  Edges are sorted by (i, j) in active numbering, which is
    // ascending global id as well.
    for (size_t t = 0; t < active.size(); ++t) {
      pre[active[t]] = merged[active[t]] == 1;  // centers at the start of the table (:110-114)
      pushed[active[t]] = 0;
    }
    uint64_t e = 0;
    for (size_t t = 0; t < active.size(); ++t) {
      const uint32_t i = active[t];
      const uint64_t e0 = e;
      while (e < n_edges && ei[e] == t) ++e;
      if (merged[i] != 0) continue;
      uint32_t absorber = 0xffffffffu;
      // the centers list is [centers at table start, ascending] ++ [candidates in push order]
      for (uint64_t x = e0; x < e && absorber == 0xffffffffu; ++x)
        if (pre[active[ej[x]]]) absorber = active[ej[x]];
      for (uint64_t x = e0; x < e && absorber == 0xffffffffu; ++x)
        if (active[ej[x]] < i && pushed[active[ej[x]]]) absorber = active[ej[x]];
      if (absorber != 0xffffffffu) {
        owner[i] = absorber;        // clusters[center].AddPoint(i)  (:121)
        merged[absorber] = 1;       // "to be the real center"        (:122)
        merged[i] = 2;              //                                 (:123)
        absorbed_table[i] = l;
      } else {
        pushed[i] = 1;              // centers.push_back(i)            (:128-130)
      }
    }
  }
  return HS_OK;
}
