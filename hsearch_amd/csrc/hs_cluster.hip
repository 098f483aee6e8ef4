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
#include <chrono>
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

struct hs_cluster_state {
  hs_params params;
  std::vector<double> a, b, coords;
  bool have_coords = false;
  const uint8_t* codes = nullptr;
  uint64_t n = 0;
  double R = 0;
  std::vector<uint8_t> merged;
  std::vector<uint32_t> owner, absorbed_table;
  // table in progress
  uint32_t table = 0xffffffffu;
  std::vector<uint32_t> active;  // not-absorbed k-mers at the start of the table, ascending
  // one L = 1 handle serves every table: new planes + rebuild (buffers, streams and events stay)
  hs_handle* h = nullptr;
  uint32_t built_table = 0xffffffffu;  // table h's index currently holds
};

namespace {

// HS_CLUSTER_TIMING=1: per-phase wall time of every table on stderr
struct PhaseTimer {
  bool on;
  std::chrono::steady_clock::time_point t;
  PhaseTimer() : on(getenv("HS_CLUSTER_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
  void lap(const char* what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "  %-12s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

// BuildLSHTalbe (hclust2.cpp:74-84): every k-mer with merged != 2, ascending id
void start_table(hs_cluster_state* st, uint32_t l) {
  if (st->table == l) return;
  st->table = l;
  st->active.clear();
  for (uint64_t i = 0; i < st->n; ++i)
    if (st->merged[i] != 2) st->active.push_back((uint32_t)i);
}

}  // namespace

extern "C" hs_status hs_clustering_begin(const hs_params* params, const double* a, const double* b,
                                         const double* coords, const uint8_t* codes, uint64_t n,
                                         double R, hs_cluster_state** out, char* err,
                                         uint32_t err_cap) {
  if (!params || !a || !b || !out || (n && !codes)) return HS_ERR_INVALID;
  *out = nullptr;
  if (n >= (1ull << 31)) {
    set_err(err, err_cap, "n must be < 2^31");
    return HS_ERR_INVALID;
  }
  hs_cluster_state* st = new hs_cluster_state;
  st->params = *params;
  const size_t d = 8 * (size_t)params->k, LK = (size_t)params->L * params->K;
  st->a.assign(a, a + LK * d);
  st->b.assign(b, b + LK);
  if (coords) {
    st->coords.assign(coords, coords + (size_t)(params->alphabet ? params->alphabet : 20) * 8);
    st->have_coords = true;
  }
  st->codes = codes;
  st->n = n;
  st->R = R;
  st->merged.assign(n, 0);  // hclust2.cpp:92-96
  st->owner.resize(n);
  for (uint64_t i = 0; i < n; ++i) st->owner[i] = (uint32_t)i;
  st->absorbed_table.assign(n, 0xffffffffu);
  *out = st;
  return HS_OK;
}

extern "C" hs_status hs_clustering_table_edges(hs_cluster_state* st, uint32_t l, uint32_t rank,
                                               uint32_t world, uint32_t* edge_i, uint32_t* edge_j,
                                               double* edge_dist, uint64_t cap, uint64_t* n_edges,
                                               char* err, uint32_t err_cap) {
  if (!st || !n_edges || !world || rank >= world || l >= st->params.L) return HS_ERR_INVALID;
  if (cap && (!edge_i || !edge_j)) return HS_ERR_INVALID;
  *n_edges = 0;
  PhaseTimer pt;
  start_table(st, l);
  pt.lap("active list");
  const uint32_t k = st->params.k, K = st->params.K;
  const size_t d = 8 * (size_t)k, na = st->active.size();
  const double* a_l = st->a.data() + (size_t)l * K * d;  // a fresh family per table (hclust2.cpp:104)
  const double* b_l = st->b.data() + (size_t)l * K;
  hs_status rc = HS_OK;
  if (!st->h) {
    hs_params p1 = st->params;
    p1.L = 1;
    rc = hs_create(&p1, a_l, b_l, st->have_coords ? st->coords.data() : nullptr, &st->h);
    if (rc != HS_OK) {
      set_err(err, err_cap, std::string("hs_create: ") + (st->h ? hs_last_error(st->h) : "no usable gfx950 device"));
      hs_destroy(st->h);
      st->h = nullptr;
      return rc;
    }
    st->built_table = 0xffffffffu;
  }
  hs_handle* h = st->h;
  if (st->built_table != l) {  // a capacity retry of the same table finds its index still there
    st->built_table = 0xffffffffu;
    rc = hs_set_planes(h, a_l, b_l);
    pt.lap("set planes");
    // the k-mer codes stay on the device across tables; the active rows are gathered there
    if (rc == HS_OK)
      rc = hs_index_build_subset(h, st->codes, st->n, na != st->n ? st->active.data() : nullptr, na);
    pt.lap("index build");
    if (rc == HS_OK) st->built_table = l;
  }
  uint64_t ne = 0;
  if (rc == HS_OK) {
    // this rank's contiguous block of the active k-mers (sizes differ by at most one)
    const uint64_t base = na / world, rem = na % world;
    const uint64_t lo = rank * base + std::min<uint64_t>(rank, rem);
    const uint64_t cnt = base + (rank < rem ? 1 : 0);
    rc = hs_self_join_range(h, lo, cnt, st->R, /*sqrt_test=*/1, edge_i, edge_j, nullptr, edge_dist, cap, &ne);
  }
  if (rc != HS_OK && rc != HS_ERR_CAPACITY)
    set_err(err, err_cap, std::string("table ") + std::to_string(l) + ": " + hs_last_error(h));
  pt.lap("self join");
  *n_edges = ne;
  if (rc == HS_OK)  // active numbering -> original k-mer numbers
    for (uint64_t e = 0; e < ne; ++e) {
      edge_i[e] = st->active[edge_i[e]];
      edge_j[e] = st->active[edge_j[e]];
    }
  return rc;
}

extern "C" hs_status hs_clustering_table_apply(hs_cluster_state* st, uint32_t l, const uint32_t* edge_i,
                                               const uint32_t* edge_j, uint64_t n_edges) {
  if (!st || l >= st->params.L || (n_edges && (!edge_i || !edge_j))) return HS_ERR_INVALID;
  PhaseTimer pt;
  start_table(st, l);
  const uint64_t n = st->n;
  // edges in (i, j) order: the greedy pass walks the active k-mers in ascending id, each with its
  // bucket mates within R in ascending id
  std::vector<uint64_t> es(n_edges);
  for (uint64_t e = 0; e < n_edges; ++e) {
    if (edge_i[e] >= n || edge_j[e] >= n) return HS_ERR_INVALID;
    es[e] = ((uint64_t)edge_i[e] << 32) | edge_j[e];
  }
  // one rank's list arrives in (i, j) order already, and so does the rank-ordered concatenation
  if (!std::is_sorted(es.begin(), es.end())) std::sort(es.begin(), es.end());
  std::vector<uint8_t>& merged = st->merged;
  std::vector<uint8_t> pre(n, 0), pushed(n, 0);
  for (uint32_t i : st->active) pre[i] = merged[i] == 1;  // centers at the start of the table (:110-114)
  // greedy pass (hclust2.cpp:107-132)
  uint64_t e = 0;
  for (uint32_t i : st->active) {
    while (e < n_edges && (uint32_t)(es[e] >> 32) < i) ++e;
    const uint64_t e0 = e;
    while (e < n_edges && (uint32_t)(es[e] >> 32) == i) ++e;
    if (merged[i] != 0) continue;
    uint32_t absorber = 0xffffffffu;
    // the centers list is [centers at table start, ascending] ++ [candidates in push order]
    for (uint64_t x = e0; x < e && absorber == 0xffffffffu; ++x)
      if (pre[(uint32_t)es[x]]) absorber = (uint32_t)es[x];
    for (uint64_t x = e0; x < e && absorber == 0xffffffffu; ++x)
      if ((uint32_t)es[x] < i && pushed[(uint32_t)es[x]]) absorber = (uint32_t)es[x];
    if (absorber != 0xffffffffu) {
      st->owner[i] = absorber;        // clusters[center].AddPoint(i)  (:121)
      merged[absorber] = 1;           // "to be the real center"        (:122)
      merged[i] = 2;                  //                                 (:123)
      st->absorbed_table[i] = l;
    } else {
      pushed[i] = 1;                  // centers.push_back(i)            (:128-130)
    }
  }
  pt.lap("greedy pass");
  st->table = 0xffffffffu;  // the next table starts from the new merged[]
  st->built_table = 0xffffffffu;
  return HS_OK;
}

extern "C" hs_status hs_clustering_end(hs_cluster_state* st, uint8_t* merged, uint32_t* owner,
                                       uint32_t* absorbed_table) {
  if (!st) return HS_ERR_INVALID;
  if (merged && owner && absorbed_table && st->n) {
    memcpy(merged, st->merged.data(), st->n);
    memcpy(owner, st->owner.data(), st->n * 4);
    memcpy(absorbed_table, st->absorbed_table.data(), st->n * 4);
  }
  hs_destroy(st->h);
  delete st;
  return HS_OK;
}

extern "C" hs_status hs_clustering(const hs_params* params, const double* a, const double* b,
                                   const double* coords, const uint8_t* codes, uint64_t n, double R,
                                   uint8_t* merged, uint32_t* owner, uint32_t* absorbed_table,
                                   char* err, uint32_t err_cap) {
  if (!params || !a || !b || (n && (!codes || !merged || !owner || !absorbed_table))) return HS_ERR_INVALID;
  hs_cluster_state* st = nullptr;
  hs_status rc = hs_clustering_begin(params, a, b, coords, codes, n, R, &st, err, err_cap);
  if (rc != HS_OK) return rc;
  std::vector<uint32_t> ei, ej;
  std::vector<double> ed;
  for (uint32_t l = 0; l < params->L && rc == HS_OK; ++l) {
    uint64_t n_edges = 0, cap = std::max<uint64_t>(ei.size(), 4 * n + 1024);
    for (;;) {
      ei.resize(cap);
      ej.resize(cap);
      ed.resize(cap);
      rc = hs_clustering_table_edges(st, l, 0, 1, ei.data(), ej.data(), ed.data(), cap, &n_edges, err, err_cap);
      if (rc == HS_ERR_CAPACITY) {
        cap = n_edges;
        continue;
      }
      break;
    }
    if (rc == HS_OK) rc = hs_clustering_table_apply(st, l, ei.data(), ej.data(), n_edges);
  }
  if (rc != HS_OK) {
    hs_clustering_end(st, nullptr, nullptr, nullptr);
    return rc;
  }
  return hs_clustering_end(st, merged, owner, absorbed_table);
}
