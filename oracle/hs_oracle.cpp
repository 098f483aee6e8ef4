// hs_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see hs_oracle.h).
//
// CPU restatement of the reference hot path of acgtun/hsearch (hclust/src/hclust), written from
// the algorithm, with the reference's cost structure where bench.py times it as the "port" CPU
// baseline: one heap vector<double> per point, unordered_map<string, vector<uint32_t>> per table
// keyed by concatenated decimal bucket ints, an N-int memset per query, strictly sequential fp64
// with separate multiply and add (built -ffp-contract=off; the reference is built ISO C++11 -O3
// without -march, hclust/src/hclust/Makefile:33, which also never contracts).
//
// Every function cites the reference file:line it follows.  Parity: PINNED against oracle/_ref and
// tests/golden/ (see hs_oracle.h).
#include "hs_oracle.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <random>
#include <string>
#include <unordered_map>
#include <thread>
#include <vector>

#include "../include/hs_tables.h"

namespace {

typedef std::unordered_map<std::string, std::vector<uint32_t> > Table;  // motif_both_points.cpp:25

// lsh.hpp:33-42 -- sum_{i<d} point[i]*a[i], left to right, product rounded, then sum rounded.
inline double dot_sequential(const double* point, const double* a, uint32_t d) {
  double acc = 0;
  for (uint32_t i = 0; i < d; ++i) acc += point[i] * a[i];
  return acc;
}

// lsh.hpp:44-49 -- int(floor((dot + b) / W)).
inline int bucket_index(double dot, double b, double W) {
  double val = dot + b;
  return (int)floor(val / W);
}

// lsh.hpp:51-59 -- to_string of each bucket int, appended with no separator.
inline void key_of(const double* point, const double* a, const double* b, uint32_t d, uint32_t K,
                   double W, std::string& key) {
  key.clear();
  for (uint32_t k = 0; k < K; ++k)
    key += std::to_string(bucket_index(dot_sequential(point, a + (size_t)k * d, d), b[k], W));
}

// motif_both_points.cpp:176-183 -- sum (a_i - b_i)^2 left to right.
inline double dist2_sequential(const double* x, const double* c, uint32_t d) {
  double dis = 0.0, r = 0.0;
  for (uint32_t i = 0; i < d; ++i) {
    r = x[i] - c[i];
    dis += r * r;
  }
  return dis;
}

}  // namespace

struct hso_index {
  uint32_t d, K, L;
  double W;
  std::vector<double> a, b;                 // [L][K][d], [L][K]
  std::vector<std::vector<double> > points;  // one heap vector per DB point, as the reference
  std::vector<Table> tables;
  std::vector<int> label;                   // motif_both_points.cpp:222
};

extern "C" {

void hso_embed_codes(const uint8_t* codes, uint64_t n, uint32_t k, double* out) {
  // hclust2.cpp:49-62: for each residue copy its AACoordinateSize coordinates.
  for (uint64_t i = 0; i < n; ++i)
    for (uint32_t p = 0; p < k; ++p)
      for (uint32_t j = 0; j < HS_AA_DIM; ++j)
        out[(i * k + p) * HS_AA_DIM + j] = HS_AA_COORDS[codes[i * k + p]][j];
}

uint64_t hso_letters_to_codes(const char* letters, uint64_t n_letters, uint8_t* codes) {
  uint64_t unknown = 0;
  for (uint64_t i = 0; i < n_letters; ++i) {
    int c = letters[i] - 'A';
    int code = (c >= 0 && c < 26) ? HS_LETTER_TO_CODE[c] : -1;  // util.hpp:92
    if (code < 0) {
      ++unknown;
      codes[i] = 255;
    } else {
      codes[i] = (uint8_t)code;
    }
  }
  return unknown;
}

void hso_hash(const double* a, const double* b, uint32_t d, uint32_t K, double W, const double* pts,
              uint64_t n, double* dots_out, int32_t* buckets_out) {
  for (uint64_t i = 0; i < n; ++i)
    for (uint32_t k = 0; k < K; ++k) {
      double dot = dot_sequential(pts + i * d, a + (size_t)k * d, d);
      if (dots_out) dots_out[i * K + k] = dot;
      buckets_out[i * K + k] = bucket_index(dot, b[k], W);
    }
}

uint32_t hso_key_string(const int32_t* buckets, uint32_t K, char* out, uint32_t cap) {
  std::string key;
  for (uint32_t k = 0; k < K; ++k) key += std::to_string(buckets[k]);
  if (cap) {
    strncpy(out, key.c_str(), cap - 1);
    out[cap - 1] = 0;
  }
  return (uint32_t)key.size();
}

hso_index* hso_index_build(const double* a, const double* b, uint32_t d, uint32_t K, uint32_t L,
                           double W, const double* db, uint64_t n) {
  hso_index* ix = new hso_index();
  ix->d = d;
  ix->K = K;
  ix->L = L;
  ix->W = W;
  ix->a.assign(a, a + (size_t)L * K * d);
  ix->b.assign(b, b + (size_t)L * K);
  ix->points.resize(n);
  for (uint64_t i = 0; i < n; ++i) ix->points[i].assign(db + i * d, db + (i + 1) * d);
  ix->tables.resize(L);
  std::string key;
  // motif_both_points.cpp:212-218: table-major, ids ascending inside a bucket.
  for (uint32_t l = 0; l < L; ++l)
    for (uint64_t i = 0; i < n; ++i) {
      key_of(ix->points[i].data(), &ix->a[(size_t)l * K * d], &ix->b[(size_t)l * K], d, K, W, key);
      ix->tables[l][key].push_back((uint32_t)i);
    }
  ix->label.assign(n, 0);
  return ix;
}

void hso_index_free(hso_index* ix) { delete ix; }

uint64_t hso_index_table_size(const hso_index* ix, uint32_t l) { return ix->tables[l].size(); }

uint64_t hso_index_query(hso_index* ix, const double* centers, uint64_t nq, double R,
                         uint32_t* hit_q, uint32_t* hit_id, uint32_t* hit_table, double* hit_dist,
                         uint64_t cap, uint64_t* cand_out) {
  const uint32_t d = ix->d, K = ix->K, L = ix->L;
  const double R2 = R * R;  // motif_both_points.cpp:204
  uint64_t n_hits = 0;
  std::string key;
  for (uint64_t q = 0; q < nq; ++q) {
    const double* c = centers + q * d;
    if (!ix->label.empty())
      memset(&ix->label[0], 0, sizeof(int) * ix->label.size());  // :225
    for (uint32_t l = 0; l < L; ++l) {
      key_of(c, &ix->a[(size_t)l * K * d], &ix->b[(size_t)l * K], d, K, ix->W, key);  // :227
      Table::iterator it = ix->tables[l].find(key);
      if (cand_out) cand_out[q * L + l] = (it == ix->tables[l].end()) ? 0 : it->second.size();
      if (it == ix->tables[l].end()) continue;
      const std::vector<uint32_t>& ids = it->second;
      for (size_t j = 0; j < ids.size(); ++j) {
        if (ix->label[ids[j]] != 0) continue;  // :233 first-seen dedupe
        double d2 = dist2_sequential(ix->points[ids[j]].data(), c, d);
        ix->label[ids[j]] = 1;
        if (d2 <= R2) {  // :239
          if (n_hits < cap) {
            hit_q[n_hits] = (uint32_t)q;
            hit_id[n_hits] = ids[j];
            hit_table[n_hits] = l;
            hit_dist[n_hits] = sqrt(d2);  // :241
          }
          ++n_hits;
        }
      }
    }
  }
  return n_hits;
}

// EXTENSION (not in the reference, which is single-threaded): the same query loop with the queries
// split into contiguous blocks over `threads` host threads, each with its own label[] array; hits
// are concatenated in block order, i.e. in the single-threaded order.  For the CPU baseline's
// "all host cores" figure only; results are identical to hso_index_query.
uint64_t hso_index_query_mt(hso_index* ix, const double* centers, uint64_t nq, double R, uint32_t threads,
                            uint32_t* hit_q, uint32_t* hit_id, uint32_t* hit_table, double* hit_dist,
                            uint64_t cap) {
  const uint32_t d = ix->d, K = ix->K, L = ix->L;
  const double R2 = R * R;
  if (threads < 1) threads = 1;
  struct Hit { uint32_t q, id, table; double dist; };
  std::vector<std::vector<Hit>> part(threads);
  std::vector<std::thread> pool;
  for (uint32_t t = 0; t < threads; ++t)
    pool.emplace_back([&, t]() {
      const uint64_t lo = nq * t / threads, hi = nq * (t + 1) / threads;
      std::vector<int> label(ix->points.size(), 0);
      std::string key;
      for (uint64_t q = lo; q < hi; ++q) {
        const double* c = centers + q * d;
        if (!label.empty()) memset(&label[0], 0, sizeof(int) * label.size());  // :225
        for (uint32_t l = 0; l < L; ++l) {
          key_of(c, &ix->a[(size_t)l * K * d], &ix->b[(size_t)l * K], d, K, ix->W, key);
          Table::const_iterator it = ix->tables[l].find(key);
          if (it == ix->tables[l].end()) continue;
          const std::vector<uint32_t>& ids = it->second;
          for (size_t j = 0; j < ids.size(); ++j) {
            if (label[ids[j]] != 0) continue;
            const double d2 = dist2_sequential(ix->points[ids[j]].data(), c, d);
            label[ids[j]] = 1;
            if (d2 <= R2) part[t].push_back(Hit{(uint32_t)q, ids[j], l, sqrt(d2)});
          }
        }
      }
    });
  for (auto& th : pool) th.join();
  uint64_t n_hits = 0;
  for (uint32_t t = 0; t < threads; ++t)
    for (const Hit& hh : part[t]) {
      if (n_hits < cap) {
        hit_q[n_hits] = hh.q;
        hit_id[n_hits] = hh.id;
        hit_table[n_hits] = hh.table;
        hit_dist[n_hits] = hh.dist;
      }
      ++n_hits;
    }
  return n_hits;
}

int hso_write_hits(const char* path, const uint32_t* hit_q, const uint32_t* hit_id,
                   const double* hit_dist, uint64_t n_hits, const char* const* q_names,
                   const char* const* db_names) {
  std::ofstream fout(path);
  if (!fout) return -1;
  for (uint64_t i = 0; i < n_hits; ++i) {
    // motif_both_points.cpp:240-241, operator<<(double) at default precision.
    if (q_names) fout << q_names[hit_q[i]]; else fout << hit_q[i];
    fout << " ";
    if (db_names) fout << db_names[hit_id[i]]; else fout << hit_id[i];
    fout << " " << hit_dist[i] << std::endl;
  }
  return 0;
}

void hso_pairwise_square(const double* db, uint64_t n, const double* centers, uint64_t nq,
                         uint32_t d, double* out) {
  for (uint64_t q = 0; q < nq; ++q)
    for (uint64_t j = 0; j < n; ++j) out[q * n + j] = dist2_sequential(db + j * d, centers + q * d, d);
}

uint64_t hso_bruteforce(const double* db, uint64_t n, const double* centers, uint64_t nq,
                        uint32_t d, double R, uint32_t* hit_q, uint32_t* hit_id, double* hit_dist,
                        uint64_t cap) {
  uint64_t n_hits = 0;
  for (uint64_t q = 0; q < nq; ++q)
    for (uint64_t j = 0; j < n; ++j) {
      // motif_both_points_noLSH.cpp:27-34,44-50: sqrt form, "dis > R" goes to the non-hits file.
      double dis = sqrt(dist2_sequential(db + j * d, centers + q * d, d));
      if (dis > R) continue;
      if (n_hits < cap) {
        hit_q[n_hits] = (uint32_t)q;
        hit_id[n_hits] = (uint32_t)j;
        hit_dist[n_hits] = dis;
      }
      ++n_hits;
    }
  return n_hits;
}

void hso_bruteforce_topk(const double* db, uint64_t n, const double* centers, uint64_t nq,
                         uint32_t d, uint32_t topk, uint32_t* nn_id, double* nn_dist2) {
  std::vector<std::pair<double, uint32_t> > all(n);
  for (uint64_t q = 0; q < nq; ++q) {
    for (uint64_t j = 0; j < n; ++j)
      all[j] = std::make_pair(dist2_sequential(db + j * d, centers + q * d, d), (uint32_t)j);
    uint32_t kk = (uint32_t)std::min<uint64_t>(topk, n);
    std::partial_sort(all.begin(), all.begin() + kk, all.end());  // (dist2, id): ties by lower id
    for (uint32_t t = 0; t < topk; ++t) {
      nn_id[q * topk + t] = t < kk ? all[t].second : 0xffffffffu;
      nn_dist2[q * topk + t] = t < kk ? all[t].first : INFINITY;
    }
  }
}

}  // extern "C"

namespace {

// hclust2.cpp:86-135.  clusters[i] = member ids in absorption order (first itself, :97-99).
void clustering_core(const double* a, const double* b, uint32_t d, uint32_t K, uint32_t L, double W,
                     double R, const double* pts, uint64_t n, std::vector<uint8_t>& merged,
                     std::vector<std::vector<uint32_t> >& clusters) {
  merged.assign(n, 0);
  clusters.assign(n, std::vector<uint32_t>());
  for (uint64_t i = 0; i < n; ++i) clusters[i].push_back((uint32_t)i);
  std::string key;
  for (uint32_t l = 0; l < L; ++l) {
    Table table;
    for (uint64_t i = 0; i < n; ++i) {  // BuildLSHTalbe hclust2.cpp:74-84
      if (merged[i] == 2) continue;
      key_of(pts + i * d, a + (size_t)l * K * d, b + (size_t)l * K, d, K, W, key);
      table[key].push_back((uint32_t)i);
    }
    for (Table::iterator it = table.begin(); it != table.end(); ++it) {  // :107
      const std::vector<uint32_t>& ids = it->second;
      std::vector<uint32_t> centers;
      for (size_t i = 0; i < ids.size(); ++i)
        if (merged[ids[i]] == 1) centers.push_back(ids[i]);  // :110-114
      for (size_t i = 0; i < ids.size(); ++i) {
        if (merged[ids[i]] == 0) {
          for (size_t j = 0; j < centers.size(); ++j) {
            // hclust2.cpp:64-71,119-120: sqrt form, "<= R".
            double dis = sqrt(dist2_sequential(pts + (size_t)ids[i] * d, pts + (size_t)centers[j] * d, d));
            if (dis <= R) {
              clusters[centers[j]].push_back(ids[i]);
              merged[centers[j]] = 1;
              merged[ids[i]] = 2;
              break;
            }
          }
        }
        if (merged[ids[i]] == 0) centers.push_back(ids[i]);  // :128-130
      }
    }
  }
}

// motif_both_points.cpp:67-87.  The reference exit(0)s when dis > R + 0.1; the oracle reports NaN.
double weight(double dis, double R, bool& bad) {
  if (dis > R + 0.1) {
    bad = true;
    return 0;
  }
  if (dis < 0.0000001) return 1;
  if (dis < 24) return 1;
  double w = 1 / (dis - 24);
  if (w > 1) return 1;
  if (w < 0) return 1;
  return w;
}

struct Res {
  std::string motif, protein;
  double dis;
};
bool res_less(const Res& x, const Res& y) {  // sortCMP motif_both_points.cpp:39-44
  if (x.motif == y.motif) return x.protein < y.protein;
  return x.motif < y.motif;
}
int res_cmp(const Res& x, const Res& y) {  // CompMOTIT :46-65 (its defined paths)
  if (x.motif == y.motif) {
    if (x.protein == y.protein) return 0;
    return x.protein > y.protein ? 1 : -1;
  }
  return x.motif > y.motif ? 1 : -1;
}

}  // namespace

extern "C" {

void hso_clustering(const double* a, const double* b, uint32_t d, uint32_t K, uint32_t L, double W,
                    double R, const double* pts, uint64_t n, uint8_t* merged_out,
                    uint32_t* owner_out) {
  std::vector<uint8_t> merged;
  std::vector<std::vector<uint32_t> > clusters;
  clustering_core(a, b, d, K, L, W, R, pts, n, merged, clusters);
  for (uint64_t i = 0; i < n; ++i) {
    merged_out[i] = merged[i];
    owner_out[i] = (uint32_t)i;
  }
  for (uint64_t i = 0; i < n; ++i)
    if (merged[i] != 2)
      for (size_t j = 0; j < clusters[i].size(); ++j) owner_out[clusters[i][j]] = (uint32_t)i;
}

int hso_clustering_to_file(const double* a, const double* b, uint32_t d, uint32_t K, uint32_t L,
                           double W, double R, const double* pts, uint64_t n, const char* path) {
  std::vector<uint8_t> merged;
  std::vector<std::vector<uint32_t> > clusters;
  clustering_core(a, b, d, K, L, W, R, pts, n, merged, clusters);
  std::ofstream fout(path);
  if (!fout) return -1;
  uint32_t cluster_id = 0;
  for (uint64_t i = 0; i < n; ++i) {  // hclust2.cpp:137-148
    if (merged[i] == 1 || merged[i] == 0) {
      fout << "#clusterid:" << cluster_id++ << ":size" << clusters[i].size() << std::endl;
      for (size_t j = 0; j < clusters[i].size(); ++j) fout << clusters[i][j] << std::endl;
    }
  }
  return 0;
}

double hso_evaluate(const char* ground_truth, const char* hits, double R) {
  std::vector<Res> brute, found;
  Res r;
  {
    std::ifstream fin(ground_truth);
    while (fin >> r.motif >> r.protein >> r.dis) brute.push_back(r);  // :104-107
  }
  {
    std::ifstream fin(hits);
    while (fin >> r.motif >> r.protein >> r.dis) found.push_back(r);  // :110-114
  }
  std::sort(found.begin(), found.end(), res_less);  // :115
  size_t i = 0, j = 0;
  double tp = 0.0, fn = 0.0;
  bool bad = false;
  while (i < brute.size() && j < found.size()) {  // :121-139
    int cmp = res_cmp(brute[i], found[j]);
    if (cmp == 0) {
      tp += weight(brute[i].dis, R, bad);
      ++i;
      ++j;
    } else if (cmp == 1) {
      ++j;
    } else {
      fn += weight(brute[i].dis, R, bad);
      ++i;
    }
  }
  while (i < brute.size()) {  // :140-145
    fn += weight(brute[i].dis, R, bad);
    ++i;
  }
  if (bad) return NAN;
  return tp / (tp + fn);  // :164
}

// ---- exhaustive search as a program (row a11) -----------------------------------------------------
// Search() of motif_both_points_noLSH.cpp:36-56 with its two files.
int hso_bruteforce_to_files(const double* db, uint64_t n, const double* centers, uint64_t nq, uint32_t d,
                            double R, const char* out_path, const char* const* q_names,
                            const char* const* db_names) {
  std::ofstream fnot((std::string(out_path) + "notlessthan.txt").c_str());  // :41-43
  std::ofstream fout(out_path);
  if (!fout || !fnot) return -1;
  for (uint64_t i = 0; i < nq; ++i)
    for (uint64_t j = 0; j < n; ++j) {
      double dis = sqrt(dist2_sequential(db + j * d, centers + i * d, d));  // :27-34
      std::ofstream& f = dis > R ? fnot : fout;                             // :47-51
      f << q_names[i] << " " << db_names[j] << " " << dis << std::endl;
    }
  return 0;
}

// ---- evaluation tooling --------------------------------------------------------
// evaluate2.cpp as it runs (:73-95): sort by (motif, protein), write <path>sort.txt, tab-separated.
int64_t hso_sort_hits_file(const char* path) {
  std::ifstream fin(path);
  if (!fin) return -1;
  std::vector<Res> rec;
  Res r;
  while (fin >> r.motif >> r.protein >> r.dis) rec.push_back(r);  // :80-82
  std::sort(rec.begin(), rec.end(), res_less);                    // :88
  std::ofstream fout((std::string(path) + "sort.txt").c_str());
  for (size_t i = 0; i < rec.size(); ++i)
    fout << rec[i].motif << "\t" << rec[i].protein << "\t" << rec[i].dis << std::endl;  // :91-93
  return (int64_t)rec.size();
}

double hso_evaluate2_weight(double dis) {  // evaluate2.cpp:62-71
  if (dis > 49.38) {
    double w = dis / (2 * 49.38);
    if (w > 1) return 1;
    return dis / (2 * 49.38);
  }
  return 1 - dis / (2 * 49.38);
}

// The comparison of evaluate2.cpp:98-153 (behind the early return of :95 in the reference).
double hso_evaluate2(const char* ground_truth, const char* hits, double* tp_out, double* fn_out) {
  std::vector<Res> brute, found;
  Res r;
  {
    std::ifstream fin(ground_truth);
    while (fin >> r.motif >> r.protein >> r.dis) brute.push_back(r);
  }
  {
    std::ifstream fin(hits);
    while (fin >> r.motif >> r.protein >> r.dis) found.push_back(r);
  }
  std::sort(brute.begin(), brute.end(), res_less);
  std::sort(found.begin(), found.end(), res_less);
  size_t i = 0, j = 0;
  double tp = 0.0, fn = 0.0;
  while (i < brute.size() && j < found.size()) {
    int cmp = res_cmp(brute[i], found[j]);
    if (cmp == 0) {
      tp += hso_evaluate2_weight(brute[i].dis);
      ++i;
      ++j;
    } else if (cmp == 1) {
      ++j;
    } else {
      fn += hso_evaluate2_weight(brute[i].dis);
      ++i;
    }
  }
  for (; i < brute.size(); ++i) fn += hso_evaluate2_weight(brute[i].dis);
  if (tp_out) *tp_out = tp;
  if (fn_out) *fn_out = fn;
  return tp / (tp + fn);
}

// ---- motif families -> centroids (centerDistanceSmapling.cpp) -------------------------------------
// Center() :67-78 over KmerToCoordinates :41-56 of the members; members of family f are
// codes[first[f] .. first[f+1])[k] (rows of the coordinate table).  centers[n_families][8k].
void hso_family_centers(const uint8_t* codes, const uint32_t* first, uint32_t n_families, uint32_t k,
                        double* centers) {
  const uint32_t d = 8 * k;
  std::vector<double> pt(d);
  for (uint32_t f = 0; f < n_families; ++f) {
    double* c = centers + (size_t)f * d;
    for (uint32_t j = 0; j < d; ++j) c[j] = 0.0;
    for (uint32_t m = first[f]; m < first[f + 1]; ++m) {
      hso_embed_codes(codes + (size_t)m * k, 1, k, pt.data());
      for (uint32_t j = 0; j < d; ++j) c[j] += pt[j];          // :69-73
    }
    for (uint32_t j = 0; j < d; ++j) c[j] /= (first[f + 1] - first[f]);  // :74-76
  }
}

// A points file as cluster2datapoint() writes it (:126-134): name line, then the coordinates.
int hso_write_points_file(const char* path, const char* const* names, const double* pts, uint64_t n,
                          uint32_t d) {
  std::ofstream fout(path);
  if (!fout) return -1;
  for (uint64_t p = 0; p < n; ++p) {
    fout << names[p] << std::endl;
    fout << pts[p * d];
    for (uint32_t q = 1; q < d; ++q) fout << " " << pts[p * d + q];
    fout << std::endl;
  }
  return 0;
}

// sequencedatabase2centers() (:138-190): centre-centre distances (i < j) to inner_path, then the
// distance of every db point (the caller passes the first min(100000, N)) to every centre,
// centre-major, to random_path.  PairwiseDistance :58-65 (sqrt form).
int hso_center_sampling(const double* db, uint64_t n, const double* centers, uint64_t nc, uint32_t d,
                        const char* inner_path, const char* random_path) {
  std::ofstream fcenter(inner_path);
  if (!fcenter) return -1;
  for (uint64_t i = 0; i < nc; ++i)
    for (uint64_t j = i + 1; j < nc; ++j)
      fcenter << sqrt(dist2_sequential(centers + i * d, centers + j * d, d)) << std::endl;  // :155-159
  std::ofstream fout(random_path);
  if (!fout) return -1;
  for (uint64_t p = 0; p < nc; ++p)
    for (uint64_t q = 0; q < n; ++q)
      fout << sqrt(dist2_sequential(db + q * d, centers + p * d, d)) << std::endl;  // :178-182
  return 0;
}

// ---- Kernel-LSH pre-grouping (pcluster) -----------------------------------------------------------
void hso_klsh_draw_planes(uint32_t feat, uint32_t bits, double sigma, double* w, double* b, double* t) {
  // member order of KLSH (lsh.hpp:37-49): the distributions, then the default-seeded engine
  std::normal_distribution<double> normal(0.0, sigma * sigma);          // lsh.cpp:22
  std::uniform_real_distribution<double> uniform_1(-1.0, 1.0);          // :23
  std::uniform_real_distribution<double> uniform_pi(0.0, 2.0 * M_PI);   // :24
  std::default_random_engine generator;                                 // lsh.hpp:49
  for (uint32_t i = 0; i < bits; ++i) {                                 // :28-37
    t[i] = uniform_1(generator);
    b[i] = uniform_pi(generator);
    for (uint32_t j = 0; j < feat; ++j) w[(size_t)i * feat + j] = normal(generator);
  }
}

void hso_klsh_features(const uint8_t* classes, uint64_t len, double* feat) {
  for (uint32_t j = 0; j < 512; ++j) feat[j] = 0.0;
  for (uint64_t i = 0; i + 3 <= len; ++i)  // pcluster.cpp:27-29, Kmer2Integer util.hpp:244-250
    feat[classes[i] + 8u * classes[i + 1] + 64u * classes[i + 2]] += 1.0;
}

uint64_t hso_klsh_hash(const double* w, const double* b, const double* t, uint32_t feat, uint32_t bits,
                       const double* p) {
  uint64_t hash_value = 0;
  for (uint32_t i = 0; i < bits; ++i) {
    double sum = 0;  // Dot, lsh.cpp:8-15
    for (uint32_t j = 0; j < feat; ++j) sum += p[j] * w[(size_t)i * feat + j];
    sum = sum + b[i];  // :44
    hash_value |= (uint64_t)((std::cos(sum) + t[i]) >= 0 ? 1 : 0) << i;  // :45-46
  }
  return hash_value;
}

}  // extern "C"
