// hs_host.cpp -- see hs_host.hpp.  Host logic only: parsing, formatting, planes, evaluation; every
// number on the search path is computed on the GPU behind include/hsearch.h.
#include "hs_host.hpp"

#include <errno.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>

#include <algorithm>
#include <fstream>
#include <functional>
#include <limits>
#include <map>
#include <random>
#include <sstream>
#include <thread>
#include <unordered_map>
#include <unordered_set>

#include "../../include/hs_tables.h"
#include "../../include/hsearch.h"
#include "../../include/hsearch_dist.h"

namespace hsearch {

Planes DrawPlanes(uint32_t dim, uint32_t K, uint32_t L, double W, uint32_t seed) {
  Planes p;
  p.dim = dim;
  p.K = K;
  p.L = L;
  p.W = W;
  p.a.resize((size_t)L * K * dim);
  p.b.resize((size_t)L * K);
  for (uint32_t l = 0; l < L; ++l) {
    // one engine and one pair of distributions per table, as one LSH object owns them
    std::default_random_engine generator(seed + l);
    std::normal_distribution<double> normal(0.0, 1.0);
    std::uniform_real_distribution<double> width(0, W);
    for (uint32_t k = 0; k < K; ++k) {
      double* row = &p.a[((size_t)l * K + k) * dim];
      for (uint32_t i = 0; i < dim; ++i) row[i] = normal(generator);
      p.b[(size_t)l * K + k] = width(generator);
    }
  }
  return p;
}

bool ReadPointsFile(const std::string& path, uint32_t dim, std::vector<std::string>* names,
                    std::vector<Point>* points) {
  std::ifstream fin(path.c_str());
  if (!fin) return false;
  std::string line;
  while (std::getline(fin, line)) {
    names->push_back(line);
    std::getline(fin, line);
    std::istringstream iss(line);
    Point point;
    point.data.assign(dim, 0.0);
    for (uint32_t i = 0; i < dim; ++i) iss >> point.data[i];
    points->push_back(point);
  }
  return true;
}

bool PointsToCodes(const std::vector<Point>& pts, uint32_t dim, std::vector<double>* table,
                   std::vector<uint8_t>* codes, std::string* err) {
  const uint32_t k = dim / 8;
  table->clear();
  codes->assign(pts.size() * (size_t)k, 0);
  struct Row {
    uint64_t w[8];
    bool operator<(const Row& o) const { return memcmp(w, o.w, sizeof(w)) < 0; }
  };
  std::map<Row, uint8_t> rows;  // exact bit patterns: the GPU must hash the very same doubles
  for (size_t i = 0; i < pts.size(); ++i) {
    if (pts[i].data.size() != dim) {
      if (err) *err = "DB point with the wrong dimension";
      return false;
    }
    for (uint32_t p = 0; p < k; ++p) {
      Row r;
      memcpy(r.w, &pts[i].data[8 * p], 64);
      std::map<Row, uint8_t>::iterator it = rows.find(r);
      if (it == rows.end()) {
        if (rows.size() >= 32) {
          if (err)
            *err = "DB points are not embeddings over an alphabet of <= 32 residues; the GPU index "
                   "stores the DB as residue codes (arbitrary-point DBs are not supported)";
          return false;
        }
        const uint8_t code = (uint8_t)rows.size();
        it = rows.insert(std::make_pair(r, code)).first;
        table->insert(table->end(), &pts[i].data[8 * p], &pts[i].data[8 * p] + 8);
      }
      (*codes)[i * k + p] = it->second;
    }
  }
  if (table->empty()) table->assign(8, 0.0);
  return true;
}

namespace {

struct SearchHits {
  std::vector<uint32_t> q, id, table;
  std::vector<double> dist;
  uint64_t n = 0;
};
struct HandleCloser {
  hs_handle* h;
  ~HandleCloser() { hs_destroy(h); }
};

// how the rank threads exchange hits (SetShardTransport) and what a rank holds (SetShardPartition)
ShardTransport g_shard_transport = kTransportRccl;
ShardPartition g_shard_partition = kPartitionQueries;

// Cost of every table for the table-partitioned layout, from a sample of the DB's k-mers: the sum over the
// table's buckets of (sample k-mers in the bucket)^2 -- for queries distributed like the DB, proportional to
// the (member, query) pairs the table contributes to the join.  One short-lived handle with all L planes (no
// index) hashes the sample; buckets are told apart by the fingerprint of their K ints (a collision would only
// nudge an estimate).  Empty on any failure: the caller then deals the tables round robin.
std::vector<double> TableCosts(hs_params prm, const Planes& planes, const double* coords, const uint8_t* sample,
                               uint64_t n_sample, int device) {
  std::vector<double> cost;
  if (!sample || n_sample < 64) return cost;
  hs_handle* h = nullptr;
  prm.device = device;
  if (hs_create(&prm, planes.a.data(), planes.b.data(), coords, &h) != HS_OK) {
    hs_destroy(h);
    return cost;
  }
  HandleCloser closer = {h};
  std::vector<int32_t> ints((size_t)n_sample * prm.L * prm.K);
  if (hs_hash_codes(h, sample, n_sample, ints.data()) != HS_OK) return cost;
  cost.assign(prm.L, 0.0);
  std::vector<uint64_t> fp(n_sample);
  for (uint32_t l = 0; l < prm.L; ++l) {
    for (uint64_t i = 0; i < n_sample; ++i) fp[i] = hs_key_fingerprint(&ints[((size_t)i * prm.L + l) * prm.K], prm.K, 0);
    std::sort(fp.begin(), fp.end());
    for (uint64_t i = 0, j; i < n_sample; i = j) {
      for (j = i + 1; j < n_sample && fp[j] == fp[i]; ++j) {}
      cost[l] += (double)(j - i) * (double)(j - i);
    }
  }
  return cost;
}


// The device part of Search(): index build (through `build`, on a fresh handle) and the query loop,
// on one GPU (devices.size() == 1 and !sharded: hs_query, no communicator) or query-sharded over
// several: one host thread and one handle per GPU, the index built on every GPU, rank
// r searching its contiguous block of centres, hits all-gathered over RCCL (hs_comm_query) so that
// every rank -- rank 0 writes the file -- holds all hits in the reference's order.
// qcodes != null: the centres are k-mers of the handle's coordinate table, given as residue codes
// [nq][k]; they cross PCIe as k bytes each (hs_query_codes) and `flat` is unused.
int RunSearch(hs_params prm, const Planes& planes, const double* coords,
              const std::function<hs_status(hs_handle*, uint32_t rank)>& build, const double* flat,
              const uint8_t* qcodes, uint64_t nq, double R, const std::vector<int>& devices, bool sharded,
              SearchHits* out, std::string* err, std::vector<uint64_t>* table_sizes,
              const uint8_t* db_sample = nullptr, uint64_t n_db_sample = 0) {
  const uint32_t world = (uint32_t)devices.size();
  if (!world) {
    if (err) *err = "no device given";
    return HS_ERR_INVALID;
  }
  // table partition: rank r holds the tables tabs[r] (global numbers, ascending) of ALL k-mers
  const bool by_tables = sharded && g_shard_partition == kPartitionTables;
  const bool by_buckets = sharded && g_shard_partition == kPartitionBuckets;
  std::vector<std::vector<uint32_t>> tabs(world);
  if (by_tables) {
    if (world > prm.L) {
      if (err) *err = "table partition: more GPUs than tables";
      return HS_ERR_INVALID;
    }
    const std::vector<double> cost = TableCosts(prm, planes, coords, db_sample, n_db_sample, devices[0]);
    std::vector<uint32_t> owner(prm.L, 0);
    hs_assign_tables(cost.empty() ? nullptr : cost.data(), prm.L, world, owner.data());
    for (uint32_t l = 0; l < prm.L; ++l) tabs[owner[l]].push_back(l);
    if (table_sizes) table_sizes->assign(prm.L, 0);
  }
  const size_t plane_doubles = (size_t)prm.K * 8 * prm.k;
  auto open = [&](int device, hs_handle** h, std::string* msg, uint32_t rank = 0) -> hs_status {
    hs_params p = prm;
    p.device = device;
    hs_status st;
    if (by_tables) {
      std::vector<double> a, b;
      for (uint32_t l : tabs[rank]) {
        a.insert(a.end(), planes.a.begin() + (size_t)l * plane_doubles, planes.a.begin() + (size_t)(l + 1) * plane_doubles);
        b.insert(b.end(), planes.b.begin() + (size_t)l * prm.K, planes.b.begin() + (size_t)(l + 1) * prm.K);
      }
      p.L = (uint32_t)tabs[rank].size();
      st = hs_create(&p, a.data(), b.data(), coords, h);
    } else {
      st = hs_create(&p, planes.a.data(), planes.b.data(), coords, h);
    }
    if (st != HS_OK) {
      *msg = std::string("hs_create: ") + (*h ? hs_last_error(*h) : "no usable gfx950 device");
      hs_destroy(*h);
      *h = nullptr;
    }
    return st;
  };
  auto sizes = [&](hs_handle* h) {
    hs_index_info info;
    if (table_sizes && hs_index_info_get(h, &info) == HS_OK)
      table_sizes->assign(info.n_buckets, info.n_buckets + prm.L);
  };
  if (!sharded) {
    hs_handle* h = nullptr;
    std::string msg;
    hs_status st = open(devices[0], &h, &msg);
    if (st != HS_OK) {
      if (err) *err = msg;
      return st;
    }
    HandleCloser closer = {h};
    st = build(h, 0);
    if (st != HS_OK) {
      if (err) *err = std::string("index build: ") + hs_last_error(h);
      return st;
    }
    sizes(h);
    uint64_t cap = std::max<uint64_t>(1024, 16 * nq);
    for (;;) {
      out->q.resize(cap);
      out->id.resize(cap);
      out->table.resize(cap);
      out->dist.resize(cap);
      st = qcodes ? hs_query_codes(h, qcodes, nq, R, out->q.data(), out->id.data(), out->table.data(),
                                   out->dist.data(), cap, &out->n, nullptr)
                  : hs_query(h, flat, nq, R, out->q.data(), out->id.data(), out->table.data(), out->dist.data(),
                             cap, &out->n, nullptr);
      if (st == HS_ERR_CAPACITY) {
        cap = out->n;
        continue;
      }
      break;
    }
    if (st != HS_OK && err) *err = std::string("hs_query: ") + hs_last_error(h);
    return st;
  }
  char cerr[256] = "";
  hs_comm* comm = nullptr;
  hs_status cst = hs_comm_create(g_shard_transport == kTransportLoopback ? HS_COMM_LOOPBACK : HS_COMM_RCCL_LOCAL,
                                 devices.data(), world, &comm, cerr, sizeof(cerr));
  if (cst != HS_OK) {
    if (err) *err = std::string("hs_comm_create: ") + cerr;
    return cst;
  }
  const uint64_t d = 8ull * prm.k;
  std::vector<hs_status> status(world, HS_OK);
  std::vector<std::string> msgs(world);
  std::vector<int> built(world, 0);
  auto rank_main = [&](uint32_t r) {
    hs_handle* h = nullptr;
    hs_status st = open(devices[r], &h, &msgs[r], r);
    HandleCloser closer = {h};
    if (st == HS_OK) {
      st = build(h, r);
      if (st != HS_OK) msgs[r] = std::string("index build: ") + hs_last_error(h);
    }
    if (st == HS_OK && by_tables && table_sizes) {  // (distinct slots per rank)
      hs_index_info info;
      if (hs_index_info_get(h, &info) == HS_OK)
        for (size_t i = 0; i < tabs[r].size(); ++i) (*table_sizes)[tabs[r][i]] = info.n_buckets[i];
    }
    built[r] = st == HS_OK;
    status[r] = st;
    // every rank learns whether all indexes stand before anyone enters the exchange
    hs_comm_barrier(comm, r);
    for (uint32_t x = 0; x < world; ++x)
      if (!built[x]) return;
    if (r == 0 && !by_tables) sizes(h);
    uint64_t lo = 0, hi = 0;
    hs_shard_bounds(nq, world, r, &lo, &hi);
    SearchHits mine;  // ranks other than 0 drop theirs
    SearchHits* dst = r == 0 ? out : &mine;
    uint64_t cap = std::max<uint64_t>(1024, 16 * nq);
    for (;;) {  // every rank sees the same total, so every rank repeats (or not) together
      dst->q.resize(cap);
      dst->id.resize(cap);
      dst->table.resize(cap);
      dst->dist.resize(cap);
      hs_handle* hq = h;
#ifdef HS_TEST_HOOKS
      // fault injection (test build of the programs only): this rank enters the exchange having failed
      if (const char* fr = getenv("HS_TEST_FAIL_RANK"))
        if ((uint32_t)atoi(fr) == r) hq = nullptr;
#endif
      if (by_tables)  // every rank passes ALL centres; the merged list comes back (include/hsearch_dist.h)
        st = hs_comm_query_tables(comm, r, hq, tabs[r].data(), (uint32_t)tabs[r].size(), qcodes ? nullptr : flat, qcodes,
                                  nq, R, dst->q.data(), dst->id.data(), dst->table.data(), dst->dist.data(), cap,
                                  &dst->n);
      else if (by_buckets)  // the same, every rank with the whole index and its part of the buckets
        st = hs_comm_query_buckets(comm, r, hq, qcodes ? nullptr : flat, qcodes, nq, R, dst->q.data(), dst->id.data(),
                                   dst->table.data(), dst->dist.data(), cap, &dst->n);
      else
      st = qcodes ? hs_comm_query_codes(comm, r, hq, qcodes + lo * prm.k, hi - lo, (uint32_t)lo, R, dst->q.data(),
                                        dst->id.data(), dst->table.data(), dst->dist.data(), cap, &dst->n)
                  : hs_comm_query(comm, r, hq, flat + lo * d, hi - lo, (uint32_t)lo, R, dst->q.data(),
                                  dst->id.data(), dst->table.data(), dst->dist.data(), cap, &dst->n);
      if (st == HS_ERR_CAPACITY) {
        cap = dst->n;
        continue;
      }
      break;
    }
    if (st != HS_OK) msgs[r] = std::string("hs_comm_query: ") + hs_comm_last_error(comm, r);
    status[r] = st;
  };
  std::vector<std::thread> threads;
  for (uint32_t r = 1; r < world; ++r) threads.emplace_back(rank_main, r);
  rank_main(0);
  for (std::thread& t : threads) t.join();
  hs_comm_destroy(comm);
  for (uint32_t r = 0; r < world; ++r)
    if (status[r] != HS_OK) {
      if (err) *err = "GPU " + std::to_string(devices[r]) + ": " + msgs[r];
      return status[r];
    }
  return HS_OK;
}

bool FlattenCenters(const std::vector<Point>& centers, uint32_t dim, std::vector<double>* flat, std::string* err) {
  flat->resize((size_t)centers.size() * dim);
  for (size_t i = 0; i < centers.size(); ++i) {
    if (centers[i].data.size() != dim) {
      if (err) *err = "centre with the wrong dimension";
      return false;
    }
    memcpy(&(*flat)[i * dim], centers[i].data.data(), sizeof(double) * dim);
  }
  return true;
}

}  // namespace

void SetShardTransport(ShardTransport t) { g_shard_transport = t; }
void SetShardPartition(ShardPartition p) { g_shard_partition = p; }

int Search(const std::vector<Point>& kmers, const std::vector<Point>& centers,
           const std::vector<std::string>& kmer_names, const std::vector<std::string>& center_names,
           const uint32_t& hash_K, const uint32_t& hash_L, const double& hash_W,
           const double& hash_R, const std::string& output_file, const Planes& planes, int device,
           std::string* err, std::vector<uint64_t>* table_sizes) {
  return SearchSharded(kmers, centers, kmer_names, center_names, hash_K, hash_L, hash_W, hash_R, output_file,
                       planes, std::vector<int>(1, device), false, err, table_sizes);
}

int SearchSharded(const std::vector<Point>& kmers, const std::vector<Point>& centers,
                  const std::vector<std::string>& kmer_names, const std::vector<std::string>& center_names,
                  const uint32_t& hash_K, const uint32_t& hash_L, const double& hash_W,
                  const double& hash_R, const std::string& output_file, const Planes& planes,
                  const std::vector<int>& devices, bool use_comm, std::string* err,
                  std::vector<uint64_t>* table_sizes) {
  const uint32_t dim = planes.dim;
  if (dim == 0 || dim % 8 != 0 || planes.K != hash_K || planes.L != hash_L || planes.W != hash_W) {
    if (err) *err = "planes do not match (dim, K, L, W)";
    return HS_ERR_INVALID;
  }
  std::vector<double> table;
  std::vector<uint8_t> codes;
  if (!PointsToCodes(kmers, dim, &table, &codes, err)) return HS_ERR_INVALID;
  std::vector<double> flat;
  if (!FlattenCenters(centers, dim, &flat, err)) return HS_ERR_INVALID;
  hs_params prm;
  memset(&prm, 0, sizeof(prm));
  prm.k = dim / 8;
  prm.K = hash_K;
  prm.L = hash_L;
  prm.W = hash_W;
  prm.alphabet = (uint32_t)(table.size() / 8);
  SearchHits hits;
  const int st = RunSearch(prm, planes, table.data(),
                           [&](hs_handle* h, uint32_t) { return hs_index_build(h, codes.data(), kmers.size()); },
                           flat.data(), nullptr, centers.size(), hash_R, devices, use_comm || devices.size() > 1,
                           &hits, err, table_sizes, codes.data(), std::min<uint64_t>(kmers.size(), 32768));
  if (st != HS_OK) return st;
  std::ofstream fout(output_file.c_str());
  for (uint64_t i = 0; i < hits.n; ++i)  // :240-241
    fout << center_names[hits.q[i]] << " " << kmer_names[hits.id[i]] << " " << hits.dist[i] << std::endl;
  fout.close();
  return HS_OK;
}


bool ReadProteinFasta(const std::string& path, bool ref_compat_eq_swap, ProteinDB* db) {
  std::ifstream fin(path.c_str());
  if (!fin) return false;
  db->name.clear();
  db->start.clear();
  db->residues.clear();
  db->eq_swapped = ref_compat_eq_swap;
  std::string line;
  while (std::getline(fin, line)) {  // protein.hpp:47-66
    if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
    if (line.empty()) continue;
    if (line[0] == '>') {
      db->name.push_back(line.substr(1));
      continue;
    }
    db->start.push_back(db->residues.size());
    for (size_t i = 0; i < line.size(); ++i) {
      const char c = line[i];
      int row = (c >= 'A' && c <= 'Z') ? HS_LETTER_TO_CODE[c - 'A'] : -1;
      // the reference keeps AA20[row] and embeds base[] of that letter: rows 5 and 6 trade places
      if (ref_compat_eq_swap && (row == 5 || row == 6)) row = 11 - row;
      db->residues.push_back(row < 0 ? ProteinDB::kUnknown : (uint8_t)row);
    }
  }
  db->start.push_back(db->residues.size());
  return true;
}

int SearchProteins(const ProteinDB& db, uint32_t kmer_length, const std::vector<Point>& centers,
                   const std::vector<std::string>& center_names, const uint32_t& hash_K,
                   const uint32_t& hash_L, const double& hash_W, const double& hash_R,
                   const std::string& output_file, const Planes& planes, int device, std::string* err,
                   std::vector<uint64_t>* table_sizes, uint64_t* n_windows, bool best_per_position) {
  return SearchProteinsSharded(db, kmer_length, centers, center_names, hash_K, hash_L, hash_W, hash_R,
                               output_file, planes, std::vector<int>(1, device), false, err, table_sizes,
                               n_windows, best_per_position);
}

int SearchProteinsSharded(const ProteinDB& db, uint32_t kmer_length, const std::vector<Point>& centers,
                          const std::vector<std::string>& center_names, const uint32_t& hash_K,
                          const uint32_t& hash_L, const double& hash_W, const double& hash_R,
                          const std::string& output_file, const Planes& planes,
                          const std::vector<int>& devices, bool use_comm, std::string* err,
                          std::vector<uint64_t>* table_sizes, uint64_t* n_windows, bool best_per_position,
                          const std::vector<uint8_t>* center_codes) {
  const uint32_t dim = 8 * kmer_length;
  if (center_codes && center_codes->size() != centers.size() * (size_t)kmer_length) {
    if (err) *err = "centre codes do not match the centres";
    return HS_ERR_INVALID;
  }
  if (kmer_length == 0 || planes.dim != dim || planes.K != hash_K || planes.L != hash_L ||
      planes.W != hash_W) {
    if (err) *err = "planes do not match (dim, K, L, W)";
    return HS_ERR_INVALID;
  }
  // sequences cut at unknown letters: windows never cross a cut, buffer positions stay the file's
  std::vector<uint8_t> res(db.residues);
  std::vector<uint64_t> seg;
  const size_t n_seq = db.start.empty() ? 0 : db.start.size() - 1;
  for (size_t s = 0; s < n_seq; ++s) {
    seg.push_back(db.start[s]);
    for (uint64_t p = db.start[s]; p < db.start[s + 1]; ++p)
      if (res[p] == ProteinDB::kUnknown) {
        res[p] = 0;
        seg.push_back(p);      // the unknown letter ends a segment ...
        seg.push_back(p + 1);  // ... and forms one of its own (length 1 < k: no window)
      }
  }
  seg.push_back(db.residues.size());
  if (kmer_length == 1) {  // a length-1 segment would be a window: drop the unknown letters' own
    if (err) *err = "kmer length 1 is not supported on a FASTA database";
    return HS_ERR_INVALID;
  }
  std::vector<double> flat;
  if (!FlattenCenters(centers, dim, &flat, err)) return HS_ERR_INVALID;
  hs_params prm;
  memset(&prm, 0, sizeof(prm));
  prm.k = kmer_length;
  prm.K = hash_K;
  prm.L = hash_L;
  prm.W = hash_W;
  prm.alphabet = HS_ALPHABET;
  uint64_t n_win = 0;
  std::vector<uint32_t> win_pos(res.size() + 1);  // there are fewer windows than residues
  SearchHits hits;
  const int rst = RunSearch(
      prm, planes, &HS_AA_COORDS[0][0],
      [&](hs_handle* h, uint32_t rank) {
        uint64_t nw = 0;  // every rank enumerates the same windows; rank 0 keeps their positions
        const hs_status bst = hs_index_build_windows(h, res.data(), res.size(), seg.data(), seg.size() - 1, &nw,
                                                     rank == 0 ? win_pos.data() : nullptr);
        if (rank == 0) n_win = nw;
        return bst;
      },
      flat.data(), center_codes ? center_codes->data() : nullptr, centers.size(), hash_R, devices,
      use_comm || devices.size() > 1, &hits, err, table_sizes);
  if (rst != HS_OK) return rst;
  if (n_windows) *n_windows = n_win;
  const uint64_t n_hits = hits.n;
  const std::vector<uint32_t>&hq = hits.q, &hid = hits.id, &ht = hits.table;
  const std::vector<double>& hd = hits.dist;
  // the letter whose embedding is the stored row: with the E <-> Q exchange an input E is shown as
  // Q, which is what the reference's ProteinDB stores and prints
  const char* letters = HS_CODE_TO_LETTER;
  std::ofstream fout(output_file.c_str());
  std::vector<uint64_t> order(n_hits);
  for (uint64_t i = 0; i < n_hits; ++i) order[i] = i;
  if (best_per_position) {
    // kmer_search.cpp:96-121: visit order (table, centre); `it2->second.second > dis` replaces
    std::sort(order.begin(), order.end(), [&](uint64_t x, uint64_t y) {
      if (hid[x] != hid[y]) return hid[x] < hid[y];
      if (hd[x] != hd[y]) return hd[x] < hd[y];
      if (ht[x] != ht[y]) return ht[x] < ht[y];
      return hq[x] < hq[y];
    });
    uint64_t kept = 0;
    for (uint64_t i = 0; i < n_hits; ++i)
      if (i == 0 || hid[order[i]] != hid[order[i - 1]]) order[kept++] = order[i];
    order.resize(kept);
  }
  for (uint64_t oi = 0; oi < order.size(); ++oi) {
    const uint64_t i = order[oi];
    const uint64_t pos = win_pos[hid[i]];
    // sequence of the window: largest s with start[s] <= pos
    const size_t s = (size_t)(std::upper_bound(db.start.begin(), db.start.end() - 1, pos) - db.start.begin()) - 1;
    std::string token;
    if (s < db.name.size()) {
      std::istringstream iss(db.name[s]);
      iss >> token;
    }
    std::string kmer(kmer_length, '?');
    for (uint32_t p = 0; p < kmer_length; ++p) kmer[p] = letters[db.residues[pos + p]];
    if (best_per_position)
      fout << token << "#" << s << "$" << (pos - db.start[s]) << "@" << kmer << "*" << hid[i] << " "
           << center_names[hq[i]] << " " << hd[i] << std::endl;
    else
      fout << center_names[hq[i]] << " " << token << "#" << s << "$" << (pos - db.start[s]) << "@" << kmer
           << "*" << hid[i] << " " << hd[i] << std::endl;
  }
  fout.close();
  return HS_OK;
}

int64_t Protein2Datapoints(const ProteinDB& db, uint32_t kmer_length, uint32_t num_of_protein_out,
                           const std::string& output_file, uint32_t seed) {
  std::ofstream fout(output_file.c_str());
  if (!fout) return -1;
  const size_t n_seq = db.start.empty() ? 0 : db.start.size() - 1;
  const char* letters = HS_CODE_TO_LETTER;
  std::unordered_set<std::string> seen;
  int64_t cnt = 0;
  srand(seed);
  for (size_t i = 0; i < n_seq; ++i) {
    if (i >= num_of_protein_out) break;  // :42
    const uint64_t len = db.start[i + 1] - db.start[i];
    if (len < kmer_length) continue;
    std::string name;
    if (i < db.name.size()) {
      std::istringstream iss(db.name[i]);  // :62-64
      iss >> name;
    }
    for (uint64_t j = 0; j + kmer_length <= len;) {
      const uint64_t pos = db.start[i] + j;
      std::string kmer(kmer_length, '?');
      bool known = true;
      for (uint32_t p = 0; p < kmer_length; ++p) {
        const uint8_t row = db.residues[pos + p];
        if (row == ProteinDB::kUnknown) known = false;
        else kmer[p] = letters[row];
      }
      if (!known || seen.find(kmer) != seen.end()) {  // :49-53
        j += 30 + rand() % 20;
        continue;
      }
      seen.insert(kmer);
      fout << name << "#" << i << "$" << j << "@" << kmer << "*" << cnt << std::endl;  // :65
      for (uint32_t p = 0; p < kmer_length; ++p)                                        // :22-28
        for (uint32_t c = 0; c < 8; ++c) {
          if (p || c) fout << " ";
          fout << HS_AA_COORDS[db.residues[pos + p]][c];
        }
      fout << std::endl;
      ++cnt;
      j += 30 + rand() % 20;  // :70-71
    }
  }
  return cnt;
}

bool ReadPclusterFasta(const std::string& path, uint32_t unknown_seed, PclusterDB* db) {
  std::ifstream fin(path.c_str());
  if (!fin) return false;
  db->names.clear();
  db->seqs.clear();
  std::mt19937 gen(unknown_seed);
  std::string line, sequence;
  bool open = false;
  while (std::getline(fin, line)) {  // read_proteins.cpp:13-35
    if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
    if (!line.empty() && line[0] == '>') {
      if (!sequence.empty()) {
        db->seqs.push_back(sequence);
        sequence.clear();
      }
      const size_t sp = line.find_first_of(' ');
      db->names.push_back(sp == std::string::npos ? line.substr(1) : line.substr(1, sp - 1));
      open = true;
      continue;
    }
    for (size_t i = 0; i < line.size(); ++i) {
      const char c = line[i];
      if (c >= 'A' && c <= 'Z' && HS_LETTER_TO_CODE[c - 'A'] >= 0)
        sequence.push_back(c);
      else if (isalpha((unsigned char)c))
        sequence.push_back(HS_CODE_TO_LETTER[gen() % 20]);
    }
  }
  if (!sequence.empty()) db->seqs.push_back(sequence);
  (void)open;
  return true;
}

int PreClustering(const PclusterDB& db, int device,
                  std::map<uint64_t, std::vector<uint32_t> >* hash_buckets, std::string* err) {
  hash_buckets->clear();
  const uint32_t bits = 16;  // pcluster.cpp:13-15: feature_size = 8^HASHLEN, bit_num = 16, sigma = 0.2
  std::vector<double> w((size_t)bits * HS_KLSH_FEATURES), b(bits), t(bits);
  hs_status st = hs_klsh_draw_planes(HS_KLSH_FEATURES, bits, 0.2, w.data(), b.data(), t.data());
  if (st != HS_OK) {
    if (err) *err = "hs_klsh_draw_planes failed";
    return st;
  }
  std::vector<uint8_t> classes;
  std::vector<uint64_t> start(1, 0);
  for (size_t i = 0; i < db.seqs.size(); ++i) {
    for (size_t p = 0; p < db.seqs[i].size(); ++p)
      classes.push_back((uint8_t)HS_REDUCED_CLASS[db.seqs[i][p] - 'A']);
    start.push_back(classes.size());
  }
  std::vector<uint64_t> codes(db.seqs.size());
  char msg[256] = {0};
  st = hs_klsh_codes(device, classes.data(), classes.size(), start.data(), db.seqs.size(), w.data(),
                     b.data(), t.data(), bits, codes.data(), nullptr, msg, sizeof(msg));
  if (st != HS_OK) {
    if (err) *err = std::string("hs_klsh_codes: ") + msg;
    return st;
  }
  for (size_t i = 0; i < codes.size(); ++i)
    if (codes[i] != HS_KLSH_NONE) (*hash_buckets)[codes[i]].push_back((uint32_t)i);  // pcluster.cpp:34
  return HS_OK;
}

bool ReadKmerFasta(const std::string& path, std::vector<Kmer>* kmers) {
  std::ifstream fin(path.c_str());
  if (!fin) return false;
  std::string tok;
  while (fin >> tok) {
    if (tok[0] == '>') {
      Kmer km;
      km.name = tok.substr(1);
      fin >> km.seq;
      kmers->push_back(km);
    }
  }
  return true;
}

bool ReadPlanesFile(const std::string& path, uint32_t dim, uint32_t K, uint32_t L, double W, Planes* planes,
                    std::string* err) {
  std::ifstream fin(path.c_str(), std::ios::binary);
  if (!fin) {
    if (err) *err = "cannot open " + path;
    return false;
  }
  planes->dim = dim;
  planes->K = K;
  planes->L = L;
  planes->W = W;
  planes->a.assign((size_t)L * K * dim, 0.0);
  planes->b.assign((size_t)L * K, 0.0);
  fin.read(reinterpret_cast<char*>(planes->a.data()), planes->a.size() * sizeof(double));
  fin.read(reinterpret_cast<char*>(planes->b.data()), planes->b.size() * sizeof(double));
  char extra;
  if (!fin || fin.read(&extra, 1)) {
    if (err) *err = path + " does not hold L*K*dim + L*K doubles for these -l/-K/-L";
    return false;
  }
  for (double v : planes->a)
    if (!std::isfinite(v)) {
      if (err) *err = path + " holds a non-finite plane coefficient";
      return false;
    }
  return true;
}

bool CentersFromKmers(const std::vector<Kmer>& kmers, uint32_t kmer_length, std::vector<std::string>* names,
                      std::vector<Point>* centers, std::string* err, std::vector<uint8_t>* codes) {
  for (const Kmer& km : kmers) {
    if (km.seq.size() != kmer_length) {
      if (err) *err = "centre " + km.name + " does not have " + std::to_string(kmer_length) + " residues";
      return false;
    }
    Point pt;
    pt.data.resize(8 * (size_t)kmer_length);
    for (uint32_t p = 0; p < kmer_length; ++p) {
      const char c = km.seq[p];
      const int row = (c >= 'A' && c <= 'Z') ? HS_LETTER_TO_CODE[c - 'A'] : -1;  // base[], util.hpp:92
      if (row < 0) {
        if (err) *err = "centre " + km.name + " has a letter outside the 20-letter alphabet";
        return false;
      }
      for (int j = 0; j < 8; ++j) pt.data[8 * p + j] = HS_AA_COORDS[row][j];  // hclust2.cpp:57-59
      if (codes) codes->push_back((uint8_t)row);
    }
    names->push_back(km.name);
    centers->push_back(pt);
  }
  return true;
}

int Clustering(const std::vector<Kmer>& kmers, const uint32_t& hash_K, const uint32_t& hash_L,
               const double& hash_W, const double& hash_R, const std::string& output_file,
               const Planes& planes, int device, uint32_t unknown_seed, std::string* err,
               uint64_t* n_clusters) {
  const uint32_t dim = planes.dim, k = dim / 8;
  if (dim == 0 || dim % 8 != 0 || planes.K != hash_K || planes.L != hash_L || planes.W != hash_W) {
    if (err) *err = "planes do not match (dim, K, L, W)";
    return HS_ERR_INVALID;
  }
  const size_t n = kmers.size();
  std::vector<uint8_t> codes(n * (size_t)k);
  std::minstd_rand unknown(unknown_seed);
  for (size_t i = 0; i < n; ++i) {
    if (kmers[i].seq.size() != k) {
      if (err) *err = "k-mer '" + kmers[i].name + "' does not have length " + std::to_string(k);
      return HS_ERR_INVALID;
    }
    for (uint32_t p = 0; p < k; ++p) {
      const int c = kmers[i].seq[p] - 'A';
      int code = (c >= 0 && c < 26) ? HS_LETTER_TO_CODE[c] : -1;  // base[], util.hpp:92
      if (code < 0) code = (int)(unknown() % 20);
      codes[i * k + p] = (uint8_t)code;
    }
  }
  hs_params prm;
  memset(&prm, 0, sizeof(prm));
  prm.k = k;
  prm.K = hash_K;
  prm.L = hash_L;
  prm.W = hash_W;
  prm.device = device;
  std::vector<uint8_t> merged(n);
  std::vector<uint32_t> owner(n), table(n);
  char msg[512] = {0};
  const hs_status st = hs_clustering(&prm, planes.a.data(), planes.b.data(), nullptr, codes.data(), n,
                                     hash_R, merged.data(), owner.data(), table.data(), msg, sizeof(msg));
  if (st != HS_OK) {
    if (err) *err = msg;
    return st;
  }
  // members of a cluster in absorption order: table by table, ascending id inside a table
  std::vector<std::vector<uint32_t> > members(n);
  std::vector<uint32_t> order;
  for (size_t i = 0; i < n; ++i)
    if (merged[i] == 2) order.push_back((uint32_t)i);
  std::stable_sort(order.begin(), order.end(),
                   [&](uint32_t x, uint32_t y) { return table[x] < table[y]; });
  for (uint32_t i : order) members[owner[i]].push_back(i);
  std::ofstream fout(output_file.c_str());
  uint32_t cluster_id = 0;
  for (size_t i = 0; i < n; ++i) {
    if (merged[i] == 1 || merged[i] == 0) {
      fout << "#clusterid:" << cluster_id++ << ":size" << members[i].size() + 1 << std::endl;
      fout << kmers[i].name << std::endl;
      for (uint32_t j : members[i]) fout << kmers[j].name << std::endl;
    }
  }
  fout.close();
  if (n_clusters) *n_clusters = cluster_id;
  return HS_OK;
}

namespace {

struct MotifRes {
  std::string motif, protein;
  double dis;
};

bool ResLess(const MotifRes& a, const MotifRes& b) {
  if (a.motif == b.motif) return a.protein < b.protein;
  return a.motif < b.motif;
}

int ResCompare(const MotifRes& a, const MotifRes& b) {
  if (a.motif == b.motif) {
    if (a.protein == b.protein) return 0;
    return a.protein > b.protein ? 1 : -1;
  }
  return a.motif > b.motif ? 1 : -1;
}

// weight(): 1 below distance 24, then 1/(dis-24) clipped to [0,1]; inconsistent ground truth
// (dis > R + 0.1) is reported through *bad instead of exit(0).
double Weight(double dis, double R, bool* bad) {
  if (dis > R + 0.1) {
    *bad = true;
    return 0;
  }
  if (dis < 0.0000001) return 1;
  if (dis < 24) return 1;
  double w = 1 / (dis - 24);
  if (w > 1) return 1;
  if (w < 0) return 1;
  return w;
}

}  // namespace

double Evaluate(const std::string& ground_truth, const std::string& output_file, const double& hash_R) {
  std::vector<MotifRes> brute, found;
  MotifRes r;
  {
    std::ifstream fin(ground_truth.c_str());
    while (fin >> r.motif >> r.protein >> r.dis) brute.push_back(r);
  }
  {
    std::ifstream fin(output_file.c_str());
    while (fin >> r.motif >> r.protein >> r.dis) found.push_back(r);
  }
  std::sort(found.begin(), found.end(), ResLess);
  size_t i = 0, j = 0;
  double tp = 0.0, fn = 0.0;
  bool bad = false;
  std::unordered_map<int, int> tp_map, fn_map;  // distance decile histogram
  while (i < brute.size() && j < found.size()) {
    const int cmp = ResCompare(brute[i], found[j]);
    if (cmp == 0) {
      tp += Weight(brute[i].dis, hash_R, &bad);
      tp_map[int(brute[i].dis * 100 / 10)]++;
      ++i;
      ++j;
    } else if (cmp == 1) {
      ++j;
    } else {
      fn += Weight(brute[i].dis, hash_R, &bad);
      fn_map[int(brute[i].dis * 100 / 10)]++;
      ++i;
    }
  }
  while (i < brute.size()) {
    fn += Weight(brute[i].dis, hash_R, &bad);
    fn_map[int(brute[i].dis * 100 / 10)]++;
    ++i;
  }
  std::ofstream fout((output_file + ".accuracy.txt").c_str());
  for (int b = 0; b < 500; ++b) {
    const bool has_fn = fn_map.find(b) != fn_map.end(), has_tp = tp_map.find(b) != tp_map.end();
    if (has_fn && has_tp)
      fout << b << " " << tp_map[b] / (fn_map[b] + (double)tp_map[b]) << " " << tp_map[b] << " "
           << fn_map[b] << std::endl;
    else if (has_fn)
      fout << b << " " << 0 << " fn " << fn_map[b] << std::endl;
    else if (has_tp)
      fout << b << " " << 1 << " tp " << tp_map[b] << std::endl;
  }
  fout.close();
  if (bad) return NAN;
  return tp / (tp + fn);
}

namespace {

// An index over `points` that only serves hs_bruteforce: one table, one function (the brute-force
// scan reads the resident residue codes, not the tables).
struct ScanEngine {
  hs_handle* h = nullptr;
  uint32_t dim = 0;
  ~ScanEngine() { hs_destroy(h); }
  int Open(const std::vector<Point>& points, size_t n_points, uint32_t dim_, int device, std::string* err) {
    dim = dim_;
    std::vector<double> table;
    std::vector<uint8_t> codes;
    std::vector<Point> head;
    const std::vector<Point>* src = &points;
    if (n_points < points.size()) {
      head.assign(points.begin(), points.begin() + n_points);
      src = &head;
    }
    if (!PointsToCodes(*src, dim, &table, &codes, err)) return HS_ERR_INVALID;
    const Planes planes = DrawPlanes(dim, 1, 1, 1.0, 0);
    hs_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.k = dim / 8;
    prm.K = 1;
    prm.L = 1;
    prm.W = 1.0;
    prm.device = device;
    prm.alphabet = (uint32_t)(table.size() / 8);
    hs_status st = hs_create(&prm, planes.a.data(), planes.b.data(), table.data(), &h);
    if (st != HS_OK) {
      if (err) *err = std::string("hs_create: ") + (h ? hs_last_error(h) : "no usable gfx950 device");
      return st;
    }
    st = hs_index_build(h, codes.data(), src->size());
    if (st != HS_OK && err) *err = std::string("hs_index_build: ") + hs_last_error(h);
    return st;
  }
  // all (centre, point) pairs of centres [c0, c0 + nc) with !(dist > R), centre-major
  int Scan(const std::vector<Point>& centers, size_t c0, size_t nc, double R, std::vector<uint32_t>* hq,
           std::vector<uint32_t>* hid, std::vector<double>* hd, uint64_t* n_hits, std::string* err) {
    std::vector<double> flat(nc * dim);
    for (size_t i = 0; i < nc; ++i) {
      if (centers[c0 + i].data.size() != dim) {
        if (err) *err = "centre with the wrong dimension";
        return HS_ERR_INVALID;
      }
      memcpy(&flat[i * dim], centers[c0 + i].data.data(), sizeof(double) * dim);
    }
    uint64_t cap = std::max<uint64_t>(hq->size(), 1024);
    for (;;) {
      hq->resize(cap);
      hid->resize(cap);
      hd->resize(cap);
      const hs_status st = hs_bruteforce(h, flat.data(), nc, R, hq->data(), hid->data(), hd->data(), cap, n_hits);
      if (st == HS_ERR_CAPACITY) {
        cap = *n_hits;
        continue;
      }
      if (st != HS_OK && err) *err = std::string("hs_bruteforce: ") + hs_last_error(h);
      return st;
    }
  }
};

}  // namespace

int SearchBruteForce(const std::vector<Point>& kmers, const std::vector<Point>& centers,
                     const std::vector<std::string>& kmer_names,
                     const std::vector<std::string>& center_names, const double& hash_R,
                     const std::string& output_file, int device, std::string* err,
                     bool write_not_less_than) {
  if (kmers.empty() || kmers[0].data.empty() || kmers[0].data.size() % 8 != 0) {
    if (err) *err = "no database points (or dimension not a multiple of 8)";
    return HS_ERR_INVALID;
  }
  const uint32_t dim = (uint32_t)kmers[0].data.size();
  ScanEngine eng;
  int st = eng.Open(kmers, kmers.size(), dim, device, err);
  if (st != HS_OK) return st;
  std::ofstream fout(output_file.c_str());
  std::ofstream fnot;
  if (write_not_less_than) fnot.open((output_file + "notlessthan.txt").c_str());  // :41-43
  // with the second file every pair comes back (R = inf) and is split here by the reference's
  // test `dis > hash_R` (:47); centre blocks bound the host buffers to ~2^24 pairs
  const double R = write_not_less_than ? std::numeric_limits<double>::infinity() : hash_R;
  const size_t block = write_not_less_than
                           ? std::max<size_t>(1, (size_t)(1u << 24) / std::max<size_t>(1, kmers.size()))
                           : centers.size();
  std::vector<uint32_t> hq, hid;
  std::vector<double> hd;
  for (size_t c0 = 0; c0 < centers.size(); c0 += block) {
    const size_t nc = std::min(block, centers.size() - c0);
    uint64_t n_hits = 0;
    st = eng.Scan(centers, c0, nc, R, &hq, &hid, &hd, &n_hits, err);
    if (st != HS_OK) return st;
    for (uint64_t i = 0; i < n_hits; ++i) {
      std::ofstream& f = (write_not_less_than && hd[i] > hash_R) ? fnot : fout;
      f << center_names[c0 + hq[i]] << " " << kmer_names[hid[i]] << " " << hd[i] << std::endl;
    }
  }
  return HS_OK;
}

bool SortHitsFile(const std::string& hits_file, uint64_t* n_records) {
  std::ifstream fin(hits_file.c_str());
  if (!fin) return false;
  std::vector<MotifRes> rec;
  MotifRes r;
  while (fin >> r.motif >> r.protein >> r.dis) rec.push_back(r);
  std::sort(rec.begin(), rec.end(), ResLess);  // evaluate2.cpp:88
  std::ofstream fout((hits_file + "sort.txt").c_str());
  for (size_t i = 0; i < rec.size(); ++i)
    fout << rec[i].motif << "\t" << rec[i].protein << "\t" << rec[i].dis << std::endl;  // :91-93
  if (n_records) *n_records = rec.size();
  return true;
}

namespace {
// weight() of evaluate2.cpp:62-71
double Weight2(double dis) {
  if (dis > 49.38) {
    const double w = dis / (2 * 49.38);
    if (w > 1) return 1;
    return dis / (2 * 49.38);
  }
  return 1 - dis / (2 * 49.38);
}
}  // namespace

double Evaluate2(const std::string& ground_truth, const std::string& hits_file, double* tp_out,
                 double* fn_out) {
  std::vector<MotifRes> brute, found;
  MotifRes r;
  {
    std::ifstream fin(ground_truth.c_str());
    while (fin >> r.motif >> r.protein >> r.dis) brute.push_back(r);
  }
  {
    std::ifstream fin(hits_file.c_str());
    while (fin >> r.motif >> r.protein >> r.dis) found.push_back(r);
  }
  std::sort(brute.begin(), brute.end(), ResLess);   // :88
  std::sort(found.begin(), found.end(), ResLess);   // :127
  size_t i = 0, j = 0;
  double tp = 0.0, fn = 0.0;
  while (i < brute.size() && j < found.size()) {  // :131-143
    const int cmp = ResCompare(brute[i], found[j]);
    if (cmp == 0) {
      tp += Weight2(brute[i].dis);
      ++i;
      ++j;
    } else if (cmp == 1) {
      ++j;
    } else {
      fn += Weight2(brute[i].dis);
      ++i;
    }
  }
  for (; i < brute.size(); ++i) fn += Weight2(brute[i].dis);  // :144-147
  if (tp_out) *tp_out = tp;
  if (fn_out) *fn_out = fn;
  return tp / (tp + fn);
}

bool ReadMotifFamilies(const std::string& path, uint32_t min_size, std::vector<MotifFamily>* families) {
  std::ifstream fin(path.c_str());
  if (!fin) return false;
  families->clear();
  MotifFamily cur;
  std::string line;
  while (std::getline(fin, line)) {  // centerDistanceSmapling.cpp:443-453
    if (line.size() == 0) continue;
    if (line[0] == '#') {
      if (cur.seqs.size() >= min_size) families->push_back(cur);
      cur.name = line;
      cur.seqs.clear();
    } else {
      cur.seqs.push_back(line);
    }
  }
  if (cur.seqs.size() >= min_size) families->push_back(cur);  // :455-457
  return true;
}

bool FamilyCenters(const std::vector<MotifFamily>& families, uint32_t kmer_length,
                   std::vector<Point>* centers, std::string* err) {
  const uint32_t dim = 8 * kmer_length;
  centers->clear();
  for (size_t f = 0; f < families.size(); ++f) {
    Point center;
    center.data.assign(dim, 0.0);
    const std::vector<std::string>& seqs = families[f].seqs;
    for (size_t m = 0; m < seqs.size(); ++m) {
      if (seqs[m].size() != kmer_length) {
        if (err) *err = "family " + families[f].name + ": member '" + seqs[m] + "' is not a " +
                        std::to_string(kmer_length) + "-mer";
        return false;
      }
      for (uint32_t p = 0; p < kmer_length; ++p) {
        const char c = seqs[m][p];
        const int row = (c >= 'A' && c <= 'Z') ? HS_LETTER_TO_CODE[c - 'A'] : -1;
        if (row < 0) {
          if (err) *err = "family " + families[f].name + ": letter outside the 20-letter alphabet in '" + seqs[m] + "'";
          return false;
        }
        for (uint32_t j = 0; j < 8; ++j) center.data[8 * p + j] += HS_AA_COORDS[row][j];  // Center() :69-73
      }
    }
    for (uint32_t j = 0; j < dim; ++j) center.data[j] /= seqs.size();  // :74-76
    centers->push_back(center);
  }
  return true;
}

bool Cluster2DataPoint(const std::vector<MotifFamily>& families, const std::vector<Point>& centers,
                       const std::string& output_file) {
  std::ofstream fout((output_file + "hclust.format.txt").c_str());
  if (!fout) return false;
  for (size_t p = 0; p < centers.size(); ++p) {  // :126-134
    fout << families[p].name << std::endl;
    fout << centers[p].data[0];
    for (size_t q = 1; q < centers[p].data.size(); ++q) fout << " " << centers[p].data[q];
    fout << std::endl;
  }
  return true;
}

int SequenceDatabase2Centers(const std::vector<Point>& kmers_proteins,
                             const std::vector<Point>& centers, const std::string& output_file,
                             const std::string& dir, int device, std::string* err) {
  if (centers.empty() || kmers_proteins.empty()) {
    if (err) *err = "no centres or no database points";
    return HS_ERR_INVALID;
  }
  const uint32_t dim = (uint32_t)centers[0].data.size();
  if (mkdir(dir.c_str(), 0777) != 0 && errno != EEXIST) {
    if (err) *err = "cannot create " + dir;
    return HS_ERR_IO;
  }
  {
    // :152-161 -- PairwiseDistance (:58-65): sequential fp64, sqrt
    std::ofstream fcenter((dir + "/" + output_file + "innercenter_protein_centers_0.txt").c_str());
    if (!fcenter) {
      if (err) *err = "cannot write into " + dir;
      return HS_ERR_IO;
    }
    for (size_t i = 0; i < centers.size(); ++i)
      for (size_t j = i + 1; j < centers.size(); ++j) {
        double dis = 0.0;
        for (uint32_t t = 0; t < dim; ++t) {
          const double r = centers[i].data[t] - centers[j].data[t];
          dis += r * r;
        }
        fcenter << sqrt(dis) << std::endl;
      }
  }
  const size_t n_sample = std::min<size_t>(100000, kmers_proteins.size());  // :166-173
  ScanEngine eng;
  int st = eng.Open(kmers_proteins, n_sample, dim, device, err);
  if (st != HS_OK) return st;
  std::ofstream fout((dir + "/" + output_file + "ramdom_protein_centers_0.txt").c_str());
  std::vector<uint32_t> hq, hid;
  std::vector<double> hd;
  const size_t block = std::max<size_t>(1, (size_t)(1u << 24) / n_sample);
  for (size_t c0 = 0; c0 < centers.size(); c0 += block) {  // :178-182, centre-major
    const size_t nc = std::min(block, centers.size() - c0);
    uint64_t n_hits = 0;
    st = eng.Scan(centers, c0, nc, std::numeric_limits<double>::infinity(), &hq, &hid, &hd, &n_hits, err);
    if (st != HS_OK) return st;
    if (n_hits != (uint64_t)nc * n_sample) {
      if (err) *err = "brute-force scan returned an incomplete distance matrix";
      return HS_ERR_STATE;
    }
    for (uint64_t i = 0; i < n_hits; ++i) fout << hd[i] << std::endl;
  }
  return HS_OK;
}

}  // namespace hsearch
