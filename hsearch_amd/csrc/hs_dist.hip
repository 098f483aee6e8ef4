// hs_dist.hip -- libhsearch_dist.so: the multi-GPU layer of include/hsearch_dist.h.
//
// Queries shard across the GPUs of a node (contiguous blocks, index replicated per GPU); the one
// exchange step of the path is a variable-length all-gather of hit tuples, done with ONE RCCL
// all-gather of per-rank records padded to the largest count:
//
//     record of rank r (m = max count, rounded up to even):  q[m] u32 | id[m] u32 | table[m] u32 | dist[m] f64
//
// Counts (and each rank's query offset) travel first: through host memory between the threads of
// one process, through a 16-byte all-gather between processes.  Packing (q made global) and
// unpacking (records -> one dense list in rank order) are device kernels on the rank's stream.
// RCCL is loaded on first use (dlopen), so the library -- and its host-memory loopback transport,
// which the CPU tests use to cover the world > 1 layout logic -- loads on machines without it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hsearch_dist.h"

namespace {

// ---- RCCL, loaded lazily --------------------------------------------------------------------------
struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {getenv("HS_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
    }
    if (!r.so) {
      r.err = std::string("cannot load RCCL: ") + dlerror();
      return;
    }
#define HS_SYM(field, name)                                              \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.so, name));      \
  if (!r.field) r.err = std::string("RCCL lacks ") + name;
    HS_SYM(GetUniqueId, "ncclGetUniqueId")
    HS_SYM(CommInitAll, "ncclCommInitAll")
    HS_SYM(CommInitRank, "ncclCommInitRank")
    HS_SYM(CommDestroy, "ncclCommDestroy")
    HS_SYM(AllGather, "ncclAllGather")
    HS_SYM(GetErrorString, "ncclGetErrorString")
#undef HS_SYM
  });
  return &r;
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct RankState {
  int device = 0;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  DevBuf send, recv, counts, io_centers, io_q, io_id, io_table, io_dist, all_q, all_id, all_table, all_dist;
  std::vector<char> host_send;  // loopback: this rank's record
  std::vector<uint32_t> h_q, h_id, h_table;  // loopback with devices: this rank's hits on the host
  std::vector<double> h_dist;
  std::vector<uint32_t> g_q, g_id, g_table;  // loopback, table partition: all ranks' tuples before the merge
  std::vector<double> g_dist;
  DevBuf table_map;                          // table partition: global numbers of the handle's tables
  std::string err;
};

// layout of one record of capacity m (m even): offsets in bytes
inline size_t rec_bytes(uint64_t m) { return (size_t)m * 20; }
inline uint64_t pad_even(uint64_t m) { return (m + 1) & ~1ull; }

__global__ __launch_bounds__(256) void hs_pack_hits_kernel(const uint32_t* __restrict__ q,
                                                           const uint32_t* __restrict__ id,
                                                           const uint32_t* __restrict__ table,
                                                           const double* __restrict__ dist, uint64_t n,
                                                           uint64_t m, uint32_t q_offset,
                                                           char* __restrict__ rec) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  uint32_t* rq = reinterpret_cast<uint32_t*>(rec);
  uint32_t* rid = rq + m;
  uint32_t* rt = rid + m;
  double* rd = reinterpret_cast<double*>(rt + m);
  const bool live = i < n;
  rq[i] = live ? q[i] + q_offset : 0u;
  rid[i] = live ? id[i] : 0u;
  rt[i] = live && table ? table[i] : 0u;
  rd[i] = live ? dist[i] : 0.0;
}

// one launch per source rank: its first `count` tuples to out[offset ...]
__global__ __launch_bounds__(256) void hs_unpack_hits_kernel(const char* __restrict__ rec, uint64_t m,
                                                             uint64_t count, uint64_t offset,
                                                             uint32_t* __restrict__ out_q,
                                                             uint32_t* __restrict__ out_id,
                                                             uint32_t* __restrict__ out_table,
                                                             double* __restrict__ out_dist) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const uint32_t* rq = reinterpret_cast<const uint32_t*>(rec);
  const uint32_t* rid = rq + m;
  const uint32_t* rt = rid + m;
  const double* rd = reinterpret_cast<const double*>(rt + m);
  out_q[offset + i] = rq[i];
  out_id[offset + i] = rid[i];
  if (out_table) out_table[offset + i] = rt[i];
  out_dist[offset + i] = rd[i];
}

// table partition: the handle numbers its tables 0 .. L_r - 1; map[l] = the table's number among all L
__global__ __launch_bounds__(256) void hs_map_tables_kernel(uint32_t* __restrict__ table, uint64_t n,
                                                            const uint32_t* __restrict__ map) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) table[i] = map[table[i]];
}

void host_pack(const uint32_t* q, const uint32_t* id, const uint32_t* table, const double* dist, uint64_t n,
               uint64_t m, uint32_t q_offset, char* rec) {
  uint32_t* rq = reinterpret_cast<uint32_t*>(rec);
  uint32_t* rid = rq + m;
  uint32_t* rt = rid + m;
  double* rd = reinterpret_cast<double*>(rt + m);
  for (uint64_t i = 0; i < m; ++i) {
    const bool live = i < n;
    rq[i] = live ? q[i] + q_offset : 0u;
    rid[i] = live ? id[i] : 0u;
    rt[i] = live && table ? table[i] : 0u;
    rd[i] = live ? dist[i] : 0.0;
  }
}

void host_unpack(const char* rec, uint64_t m, uint64_t count, uint64_t offset, uint32_t* out_q,
                 uint32_t* out_id, uint32_t* out_table, double* out_dist) {
  const uint32_t* rq = reinterpret_cast<const uint32_t*>(rec);
  const uint32_t* rid = rq + m;
  const uint32_t* rt = rid + m;
  const double* rd = reinterpret_cast<const double*>(rt + m);
  for (uint64_t i = 0; i < count; ++i) {
    out_q[offset + i] = rq[i];
    out_id[offset + i] = rid[i];
    if (out_table) out_table[offset + i] = rt[i];
    out_dist[offset + i] = rd[i];
  }
}

}  // namespace

struct hs_comm {
  int kind = HS_COMM_RCCL_LOCAL;
  bool per_process = false;  // hs_comm_create_rank: this object serves one rank of a multi-process job
  uint32_t world = 1, my_rank = 0;
  std::vector<RankState> ranks;  // [world] (per_process: [1])
  // rendezvous of the host threads of one process
  std::mutex mu;
  std::condition_variable cv;
  uint32_t arrived = 0;
  uint64_t generation = 0;
  bool has_devices = false;  // (always true for RCCL) the ranks have GPUs: hs_comm_query may run
  std::vector<uint64_t> pub;  // [world][4] words published by each rank for the current exchange
  RankState& rs(uint32_t rank) { return ranks[per_process ? 0 : rank]; }
  void barrier() {
    if (per_process || world == 1) return;
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t gen = generation;
    if (++arrived == world) {
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
  }
};

namespace {

hs_status say(char* err, uint32_t cap, hs_status st, const std::string& msg) {
  if (err && cap) {
    strncpy(err, msg.c_str(), cap - 1);
    err[cap - 1] = 0;
  }
  return st;
}

hs_status rfail(hs_comm* c, uint32_t rank, hs_status st, const std::string& msg) {
  c->rs(rank).err = msg;
  return st;
}

#define HSD_HIP(c, rank, expr)                                                                   \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return rfail(c, rank, e_ == hipErrorOutOfMemory ? HS_ERR_NOMEM : HS_ERR_HIP,               \
                   std::string(#expr) + ": " + hipGetErrorString(e_));                           \
  } while (0)
#define HSD_NCCL(c, rank, expr)                                                                  \
  do {                                                                                           \
    ncclResult_t e_ = (expr);                                                                    \
    if (e_ != ncclSuccess)                                                                       \
      return rfail(c, rank, HS_ERR_HIP, std::string(#expr) + ": " + rccl()->GetErrorString(e_)); \
  } while (0)

#define HS_DIST_CHECK(expr)           \
  do {                                \
    hs_status st_ = (expr);           \
    if (st_ != HS_OK) return st_;     \
  } while (0)

bool rank_ok(const hs_comm* c, uint32_t rank) {
  return c && rank < c->world && (!c->per_process || rank == c->my_rank);
}

}  // namespace

extern "C" {

void hs_shard_bounds(uint64_t n, uint32_t world, uint32_t rank, uint64_t* lo, uint64_t* hi) {
  if (!world) world = 1;
  const uint64_t base = n / world, rem = n % world;
  const uint64_t l = (uint64_t)rank * base + std::min<uint64_t>(rank, rem);
  if (lo) *lo = l;
  if (hi) *hi = l + base + (rank < rem ? 1 : 0);
}

hs_status hs_comm_create(int kind, const int* devices, uint32_t world, hs_comm** out, char* err,
                         uint32_t err_cap) {
  if (!out) return HS_ERR_INVALID;
  *out = nullptr;
  if (!world || world > 64 || (kind != HS_COMM_RCCL_LOCAL && kind != HS_COMM_LOOPBACK))
    return say(err, err_cap, HS_ERR_INVALID, "bad communicator kind or world size");
  hs_comm* c = new hs_comm();
  c->kind = kind;
  c->world = world;
  c->ranks.resize(world);
  c->pub.assign((size_t)world * 4, 0);
  if (kind == HS_COMM_LOOPBACK && !devices) {
    *out = c;
    return HS_OK;
  }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
    delete c;
    return say(err, err_cap, HS_ERR_NO_DEVICE, "no usable gfx950 device");
  }
  if (kind == HS_COMM_LOOPBACK) {
    // host-memory transport between ranks that HAVE devices (several ranks may share one): hs_comm_query
    // runs each rank's search on its device and exchanges the hits through host memory
    for (uint32_t r = 0; r < world; ++r) {
      if (devices[r] < 0 || devices[r] >= n_dev) {
        delete c;
        return say(err, err_cap, HS_ERR_NO_DEVICE, "device ordinal out of range");
      }
      c->ranks[r].device = devices[r];
    }
    c->has_devices = true;
    *out = c;
    return HS_OK;
  }
  c->has_devices = true;
  std::vector<int> devs(world);
  for (uint32_t r = 0; r < world; ++r) {
    devs[r] = devices ? devices[r] : (int)r;
    if (devs[r] < 0 || devs[r] >= n_dev) {
      delete c;
      return say(err, err_cap, HS_ERR_NO_DEVICE, "device ordinal out of range (a rank per GPU: --gpus must not exceed the GPUs present)");
    }
    for (uint32_t r2 = 0; r2 < r; ++r2)
      if (devs[r2] == devs[r]) {
        delete c;
        return say(err, err_cap, HS_ERR_INVALID, "two ranks on one device");
      }
    c->ranks[r].device = devs[r];
  }
  Rccl* R = rccl();
  if (!R->err.empty()) {
    delete c;
    return say(err, err_cap, HS_ERR_HIP, R->err);
  }
  std::vector<ncclComm_t> comms(world, nullptr);
  ncclResult_t e = R->CommInitAll(comms.data(), (int)world, devs.data());
  if (e != ncclSuccess) {
    delete c;
    return say(err, err_cap, HS_ERR_HIP, std::string("ncclCommInitAll: ") + R->GetErrorString(e));
  }
  for (uint32_t r = 0; r < world; ++r) {
    c->ranks[r].comm = comms[r];
    if (hipSetDevice(devs[r]) != hipSuccess ||
        hipStreamCreateWithFlags(&c->ranks[r].stream, hipStreamNonBlocking) != hipSuccess) {
      hs_comm_destroy(c);
      return say(err, err_cap, HS_ERR_HIP, "cannot create a stream on a rank's device");
    }
  }
  *out = c;
  return HS_OK;
}

hs_status hs_comm_unique_id(char id[HS_COMM_ID_BYTES]) {
  if (!id) return HS_ERR_INVALID;
  static_assert(sizeof(ncclUniqueId) <= HS_COMM_ID_BYTES, "id size");
  Rccl* R = rccl();
  if (!R->err.empty()) return HS_ERR_HIP;
  ncclUniqueId u;
  if (R->GetUniqueId(&u) != ncclSuccess) return HS_ERR_HIP;
  memset(id, 0, HS_COMM_ID_BYTES);
  memcpy(id, &u, sizeof(u));
  return HS_OK;
}

hs_status hs_comm_create_rank(const char id[HS_COMM_ID_BYTES], uint32_t rank, uint32_t world, int device,
                              hs_comm** out, char* err, uint32_t err_cap) {
  if (!out) return HS_ERR_INVALID;
  *out = nullptr;
  if (!id || !world || rank >= world) return say(err, err_cap, HS_ERR_INVALID, "bad rank / world");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
    return say(err, err_cap, HS_ERR_NO_DEVICE, "no usable gfx950 device");
  Rccl* R = rccl();
  if (!R->err.empty()) return say(err, err_cap, HS_ERR_HIP, R->err);
  hs_comm* c = new hs_comm();
  c->kind = HS_COMM_RCCL_LOCAL;
  c->per_process = true;
  c->world = world;
  c->my_rank = rank;
  c->ranks.resize(1);
  c->pub.assign((size_t)world * 4, 0);
  c->has_devices = true;
  c->ranks[0].device = device;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  if (hipSetDevice(device) != hipSuccess ||
      hipStreamCreateWithFlags(&c->ranks[0].stream, hipStreamNonBlocking) != hipSuccess ||
      c->ranks[0].counts.reserve((size_t)(world + 1) * 32) != hipSuccess) {  // phase 1 of every exchange
    hs_comm_destroy(c);
    return say(err, err_cap, HS_ERR_HIP, "cannot create a stream / the counts buffer on the rank's device");
  }
  ncclResult_t e = R->CommInitRank(&c->ranks[0].comm, (int)world, u, (int)rank);
  if (e != ncclSuccess) {
    hs_comm_destroy(c);
    return say(err, err_cap, HS_ERR_HIP, std::string("ncclCommInitRank: ") + R->GetErrorString(e));
  }
  *out = c;
  return HS_OK;
}

void hs_comm_destroy(hs_comm* c) {
  if (!c) return;
  for (RankState& r : c->ranks) {
    if (c->has_devices) {
      (void)hipSetDevice(r.device);
      if (r.stream) (void)hipStreamSynchronize(r.stream);
      DevBuf* bufs[] = {&r.send, &r.recv, &r.counts, &r.io_centers, &r.io_q, &r.io_id, &r.io_table, &r.io_dist,
                        &r.all_q, &r.all_id, &r.all_table, &r.all_dist, &r.table_map};
      for (DevBuf* b : bufs) b->release();
      if (r.comm) (void)rccl()->CommDestroy(r.comm);  // (null for the host-memory transport)
      if (r.stream) (void)hipStreamDestroy(r.stream);
    }
  }
  delete c;
}

uint32_t hs_comm_world(const hs_comm* c) { return c ? c->world : 0; }

const char* hs_comm_last_error(const hs_comm* c, uint32_t rank) {
  if (!rank_ok(c, rank)) return "bad communicator or rank";
  return const_cast<hs_comm*>(c)->rs(rank).err.c_str();
}

hs_status hs_comm_barrier(hs_comm* c, uint32_t rank) {
  if (!rank_ok(c, rank)) return HS_ERR_INVALID;
  c->barrier();
  return HS_OK;
}

// ---- the exchange -----------------------------------------------------------------------------------
// Rule of this file: once a rank has entered an exchange, every rendezvous and every collective of that
// exchange is passed by EVERY rank, whatever happened to it locally.  A rank's local failure (a bad
// argument, a failed allocation, a failed query) travels as a status word in phase 1 and every rank
// then takes the same decision from the same words: all of them stop (the failed rank with its own
// status, the others with HS_ERR_PEER), or all of them report the capacity they need, or all of them
// move data.
}  // extern "C"

namespace {

// Phase 1: four words per rank {n, q_offset, cap, status}, read by every rank.  Threads of one process:
// two rendezvous (all published, all read); one process per rank: one 32-byte all-gather through the
// buffer hs_comm_create_rank reserved.
hs_status publish_words(hs_comm* c, uint32_t rank, const uint64_t mine[4], std::vector<uint64_t>* all) {
  const uint32_t world = c->world;
  all->assign((size_t)world * 4, 0);
  if (c->per_process && world > 1) {
    RankState& me = c->rs(rank);
    Rccl* R = rccl();
    HSD_HIP(c, rank, me.counts.reserve((size_t)(world + 1) * 32));  // (reserved at creation: a no-op here)
    char* d_mine = static_cast<char*>(me.counts.p) + (size_t)world * 32;
    // Once the words are on their way this rank is IN the collective: a failed copy is remembered, the
    // all-gather is still entered (the peers are in it, or about to be), and the failure is reported after it.
    const hipError_t e_copy = hipMemcpyAsync(d_mine, mine, 32, hipMemcpyHostToDevice, me.stream);
    const ncclResult_t e_gather = R->AllGather(d_mine, me.counts.p, 32, ncclChar, me.comm, me.stream);
    HSD_HIP(c, rank, e_copy);
    HSD_NCCL(c, rank, e_gather);
    HSD_HIP(c, rank, hipMemcpyAsync(all->data(), me.counts.p, (size_t)world * 32, hipMemcpyDeviceToHost, me.stream));
    HSD_HIP(c, rank, hipStreamSynchronize(me.stream));
    return HS_OK;
  }
  memcpy(&c->pub[(size_t)rank * 4], mine, 32);
  c->barrier();  // everyone has published
  *all = c->pub;
  c->barrier();  // everyone has read: the next call may publish again
  return HS_OK;
}

// The first failed rank of a published status column, or world if none.
uint32_t first_failed(const std::vector<uint64_t>& all, uint32_t world, int col) {
  for (uint32_t r = 0; r < world; ++r)
    if (all[(size_t)r * 4 + col] != (uint64_t)HS_OK) return r;
  return world;
}

hs_status peer_failed(hs_comm* c, uint32_t rank, uint32_t r, uint64_t st, hs_status own) {
  if (own != HS_OK) return own;  // (its message is already in place)
  return rfail(c, rank, HS_ERR_PEER, "rank " + std::to_string(r) + " failed with status " + std::to_string(st) +
                                         "; nothing was exchanged");
}

// in_host: q / id / table / dist are HOST pointers (loopback); otherwise pointers of the rank's GPU.
// `lst` = what happened to this rank before the exchange (HS_OK, or the failure it carries into it).
hs_status allgather_impl(hs_comm* c, uint32_t rank, hs_status lst, const uint32_t* q, const uint32_t* id,
                         const uint32_t* table, const double* dist, uint64_t n_local, uint32_t q_offset,
                         uint32_t* out_q, uint32_t* out_id, uint32_t* out_table, double* out_dist, uint64_t cap,
                         uint64_t* n_total) {
  RankState& me = c->rs(rank);
  const uint32_t world = c->world;
  const bool loop = c->kind == HS_COMM_LOOPBACK;
  Rccl* R = loop ? nullptr : rccl();
  if (lst == HS_OK && n_local && (!q || !id || !dist)) lst = rfail(c, rank, HS_ERR_INVALID, "null hit arrays");
  if (lst == HS_OK && n_local >= (1ull << 31)) lst = rfail(c, rank, HS_ERR_INVALID, "more than 2^31 - 1 hits on one rank");
  if (lst == HS_OK && !loop && hipSetDevice(me.device) != hipSuccess)
    lst = rfail(c, rank, HS_ERR_HIP, "hipSetDevice failed on the rank's device");
  // ---- phase 1: counts, query offsets, capacities and states of all ranks
  // (a rank without output arrays offers no capacity: the decision stays one for all ranks)
  const bool outs = out_q && out_id && out_dist;
  const uint64_t mine[4] = {lst == HS_OK ? n_local : 0, q_offset, outs ? cap : 0, (uint64_t)lst};
  std::vector<uint64_t> all;
  HS_DIST_CHECK(publish_words(c, rank, mine, &all));
  uint64_t m = 0, total = 0, cap_min = ~0ull;
  std::vector<uint64_t> cnt(world);
  for (uint32_t r = 0; r < world; ++r) {
    cnt[r] = all[(size_t)r * 4];
    m = std::max(m, cnt[r]);
    total += cnt[r];
    cap_min = std::min(cap_min, all[(size_t)r * 4 + 2]);  // one decision for all ranks: the smallest capacity offered
  }
  m = pad_even(m);
  *n_total = total;
  const uint32_t bad = first_failed(all, world, 3);
  if (bad < world) return peer_failed(c, rank, bad, all[(size_t)bad * 4 + 3], lst);
  if (total > cap_min) return rfail(c, rank, HS_ERR_CAPACITY, "hit buffers too small; see *n_total");
  if (!total) return HS_OK;
  const size_t rb = rec_bytes(m);
  if (loop) {
    // ---- phase 2 (host memory): publish the record, rendezvous, copy every rank's
    me.host_send.resize(std::max<size_t>(rb, 8));
    host_pack(q, id, table, dist, n_local, m, q_offset, me.host_send.data());
    c->barrier();  // all records are in place
    uint64_t off = 0;
    for (uint32_t r = 0; r < world; ++r) {
      host_unpack(c->ranks[r].host_send.data(), m, cnt[r], off, out_q, out_id, out_table, out_dist);
      off += cnt[r];
    }
    c->barrier();  // nobody repacks while a neighbour still reads
    return HS_OK;
  }
  // ---- phase 2 (RCCL): the buffers first -- and a second, one-word round, so that a rank that cannot
  // get them keeps everybody out of the collective -- then pack, ONE all-gather, unpack
  hs_status rst = HS_OK;
  if (me.send.reserve(rb) != hipSuccess || me.recv.reserve(rb * world) != hipSuccess)
    rst = rfail(c, rank, HS_ERR_NOMEM, "no memory for the exchange buffers");
  if (rst == HS_OK) {
    // this rank's record is packed BEFORE the round below: a launch that fails travels as this rank's status,
    // and no rank unpacks a record that was never written
    const unsigned pb = (unsigned)((m + 255) / 256);
    hs_pack_hits_kernel<<<pb, 256, 0, me.stream>>>(q, id, table, dist, n_local, m, q_offset,
                                                   static_cast<char*>(me.send.p));
    const hipError_t pack_err = hipGetLastError();
    if (pack_err != hipSuccess)
      rst = rfail(c, rank, HS_ERR_HIP, std::string("hs_pack_hits_kernel: ") + hipGetErrorString(pack_err));
  }
  const uint64_t ready[4] = {(uint64_t)rst, 0, 0, 0};
  HS_DIST_CHECK(publish_words(c, rank, ready, &all));
  const uint32_t bad2 = first_failed(all, world, 0);
  if (bad2 < world) return peer_failed(c, rank, bad2, all[(size_t)bad2 * 4], rst);
  HSD_NCCL(c, rank, R->AllGather(me.send.p, me.recv.p, rb, ncclChar, me.comm, me.stream));
  uint64_t off = 0;
  for (uint32_t r = 0; r < world; ++r) {
    if (cnt[r]) {
      const unsigned ub = (unsigned)((cnt[r] + 255) / 256);
      hs_unpack_hits_kernel<<<ub, 256, 0, me.stream>>>(static_cast<const char*>(me.recv.p) + (size_t)r * rb, m,
                                                       cnt[r], off, out_q, out_id, out_table, out_dist);
      HSD_HIP(c, rank, hipGetLastError());
    }
    off += cnt[r];
  }
  HSD_HIP(c, rank, hipStreamSynchronize(me.stream));
  return HS_OK;
}

}  // namespace

extern "C" {

hs_status hs_allgather_hits(hs_comm* c, uint32_t rank, const uint32_t* q, const uint32_t* id,
                            const uint32_t* table, const double* dist, uint64_t n_local,
                            uint32_t q_offset, uint32_t* out_q, uint32_t* out_id, uint32_t* out_table,
                            double* out_dist, uint64_t cap, uint64_t* n_total) {
  // (a call that cannot even name its rank cannot take part: the one early return)
  if (!rank_ok(c, rank) || !n_total) return HS_ERR_INVALID;
  return allgather_impl(c, rank, HS_OK, q, id, table, dist, n_local, q_offset, out_q, out_id, out_table, out_dist,
                        cap, n_total);
}

}  // extern "C"

namespace {

// hs_comm_query / hs_comm_query_codes: the block as centres [nq_local][d] or as residue codes [nq_local][k]
// tables != null: the TABLE-partitioned form -- the handle holds tables[0 .. n_tables) (global numbers,
// ascending) over all k-mers, the block is ALL queries (q_offset 0), and the gathered tuples are merged
// (hs_merge_first_table_dev) before they go to the caller's arrays.  buckets: the BUCKET-partitioned form -- all
// tables, all queries, the handle restricted to the rank's part of the buckets for the call; merged the same way.
hs_status comm_query_impl(hs_comm* c, uint32_t rank, hs_handle* h, const double* centers, const uint8_t* qcodes,
                          uint64_t nq_local, uint32_t q_offset, double R, uint32_t* hit_q, uint32_t* hit_id,
                          uint32_t* hit_table, double* hit_dist, uint64_t cap, uint64_t* n_total,
                          const uint32_t* tables = nullptr, uint32_t n_tables = 0, bool buckets = false) {
  if (!rank_ok(c, rank) || !n_total) return HS_ERR_INVALID;  // cannot take part at all
  RankState& me = c->rs(rank);
  const bool loop = c->kind == HS_COMM_LOOPBACK;
  const bool merge = tables != nullptr || buckets;
  // Everything up to the exchange can fail on this rank alone: the failure is kept in `lst` and carried
  // INTO the exchange (with no hits), where every rank learns of it -- never an early return that would
  // leave the other ranks waiting at a rendezvous or inside ncclAllGather.
  hs_status lst = HS_OK;
  auto keep = [&](hs_status st, const std::string& msg) {
    if (lst == HS_OK && st != HS_OK) {
      lst = st;
      me.err = msg;
    }
  };
  auto hip = [&](hipError_t e, const char* what) {
    if (e != hipSuccess)
      keep(e == hipErrorOutOfMemory ? HS_ERR_NOMEM : HS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return e == hipSuccess;
  };
  if (!h || (nq_local && !centers && !qcodes)) keep(HS_ERR_INVALID, "null handle or centres");
  if (loop && !c->has_devices) keep(HS_ERR_INVALID, "hs_comm_query needs a communicator whose ranks have devices");
  hs_params prm;
  memset(&prm, 0, sizeof(prm));
  if (lst == HS_OK && hs_get_params(h, &prm) != HS_OK) keep(HS_ERR_INVALID, "hs_get_params failed");
  if (lst == HS_OK && prm.device != me.device) keep(HS_ERR_INVALID, "the handle is bound to another device than the rank");
  if (lst == HS_OK) hip(hipSetDevice(me.device), "hipSetDevice");
  if (lst == HS_OK && tables) {
    bool ok = n_tables == prm.L;
    for (uint32_t l = 0; ok && l < n_tables; ++l) ok = tables[l] < 32u && (l == 0 || tables[l] > tables[l - 1]);
    if (!ok) keep(HS_ERR_INVALID, "table partition: as many table numbers as the handle has tables, ascending, < 32");
    if (ok && hip(me.table_map.reserve(32 * 4), "table map: hipMalloc"))
      hip(hipMemcpy(me.table_map.p, tables, (size_t)n_tables * 4, hipMemcpyHostToDevice), "table map: hipMemcpy");
  }
  // this rank's block on its GPU, then the search with HBM-resident outputs
  const uint64_t d = 8ull * prm.k;
  const size_t cbytes = qcodes ? (size_t)nq_local * prm.k : (size_t)nq_local * d * 8;
  const void* const src = qcodes ? (const void*)qcodes : (const void*)centers;
  if (lst == HS_OK && hip(me.io_centers.reserve(std::max<size_t>(16, cbytes)), "centres: hipMalloc") && cbytes)
    hip(hipMemcpy(me.io_centers.p, src, cbytes, hipMemcpyHostToDevice), "centres: hipMemcpy");
  uint64_t lcap = std::max<uint64_t>(me.io_q.cap / 4, std::max<uint64_t>(1024, 16 * nq_local)), n_local = 0;
  if (lst == HS_OK && buckets && hs_set_bucket_partition(h, rank, (uint32_t)c->world) != HS_OK)
    keep(HS_ERR_INVALID, std::string("hs_set_bucket_partition: ") + hs_last_error(h));
  while (lst == HS_OK) {
    if (!hip(me.io_q.reserve(lcap * 4), "hits: hipMalloc") || !hip(me.io_id.reserve(lcap * 4), "hits: hipMalloc") ||
        !hip(me.io_table.reserve(lcap * 4), "hits: hipMalloc") || !hip(me.io_dist.reserve(lcap * 8), "hits: hipMalloc"))
      break;
    const hs_status st =
        qcodes ? hs_query_codes_dev(h, static_cast<const uint8_t*>(me.io_centers.p), nq_local, R,
                                    static_cast<uint32_t*>(me.io_q.p), static_cast<uint32_t*>(me.io_id.p),
                                    static_cast<uint32_t*>(me.io_table.p), static_cast<double*>(me.io_dist.p), lcap,
                                    &n_local, nullptr)
               : hs_query_dev(h, static_cast<const double*>(me.io_centers.p), nq_local, R,
                              static_cast<uint32_t*>(me.io_q.p), static_cast<uint32_t*>(me.io_id.p),
                              static_cast<uint32_t*>(me.io_table.p), static_cast<double*>(me.io_dist.p), lcap,
                              &n_local, nullptr);
    if (st == HS_ERR_CAPACITY) {
      lcap = n_local + n_local / 8 + 1024;
      continue;
    }
    if (st != HS_OK) keep(st, std::string("hs_query_dev: ") + hs_last_error(h));
    break;
  }
  if (buckets && h) (void)hs_set_bucket_partition(h, 0, 1);  // the handle answers for all buckets again
  if (lst != HS_OK) n_local = 0;
  if (tables && n_local) {
    hs_map_tables_kernel<<<(unsigned)((n_local + 255) / 256), 256, 0, me.stream>>>(
        static_cast<uint32_t*>(me.io_table.p), n_local, static_cast<const uint32_t*>(me.table_map.p));
    hip(hipGetLastError(), "hs_map_tables_kernel");
    hip(hipStreamSynchronize(me.stream), "hs_map_tables_kernel");
    if (lst != HS_OK) n_local = 0;
  }
  hs_status st;
  if (loop && merge) {
    // host-memory transport: this rank's tuples to the host, every rank's into g_*, then the merge on this
    // rank's device (through the all_* buffers) and the merged list to the caller
    me.h_q.resize(n_local);
    me.h_id.resize(n_local);
    me.h_table.resize(n_local);
    me.h_dist.resize(n_local);
    if (n_local) {
      hip(hipMemcpy(me.h_q.data(), me.io_q.p, n_local * 4, hipMemcpyDeviceToHost), "hits: hipMemcpy");
      hip(hipMemcpy(me.h_id.data(), me.io_id.p, n_local * 4, hipMemcpyDeviceToHost), "hits: hipMemcpy");
      hip(hipMemcpy(me.h_table.data(), me.io_table.p, n_local * 4, hipMemcpyDeviceToHost), "hits: hipMemcpy");
      hip(hipMemcpy(me.h_dist.data(), me.io_dist.p, n_local * 8, hipMemcpyDeviceToHost), "hits: hipMemcpy");
    }
    uint64_t gcap = std::max<uint64_t>(me.g_q.size(), 1024);
    for (;;) {
      me.g_q.resize(gcap);
      me.g_id.resize(gcap);
      me.g_table.resize(gcap);
      me.g_dist.resize(gcap);
      st = allgather_impl(c, rank, lst, me.h_q.data(), me.h_id.data(), me.h_table.data(), me.h_dist.data(),
                          lst == HS_OK ? n_local : 0, 0u, me.g_q.data(), me.g_id.data(), me.g_table.data(),
                          me.g_dist.data(), gcap, n_total);
      if (st == HS_ERR_CAPACITY) {  // every rank sees the same total and repeats the exchange
        gcap = *n_total + 1024;
        continue;
      }
      break;
    }
    if (st != HS_OK) return st;
    const uint64_t ng = *n_total;
    uint64_t kept = 0;
    if (ng) {
      HSD_HIP(c, rank, me.all_q.reserve(ng * 4));
      HSD_HIP(c, rank, me.all_id.reserve(ng * 4));
      HSD_HIP(c, rank, me.all_table.reserve(ng * 4));
      HSD_HIP(c, rank, me.all_dist.reserve(ng * 8));
      HSD_HIP(c, rank, hipMemcpy(me.all_q.p, me.g_q.data(), ng * 4, hipMemcpyHostToDevice));
      HSD_HIP(c, rank, hipMemcpy(me.all_id.p, me.g_id.data(), ng * 4, hipMemcpyHostToDevice));
      HSD_HIP(c, rank, hipMemcpy(me.all_table.p, me.g_table.data(), ng * 4, hipMemcpyHostToDevice));
      HSD_HIP(c, rank, hipMemcpy(me.all_dist.p, me.g_dist.data(), ng * 8, hipMemcpyHostToDevice));
    }
    const hs_status mst = hs_merge_first_table_dev(h, static_cast<uint32_t*>(me.all_q.p), static_cast<uint32_t*>(me.all_id.p),
                                                   static_cast<uint32_t*>(me.all_table.p), static_cast<double*>(me.all_dist.p),
                                                   ng, &kept);
    if (mst != HS_OK) return rfail(c, rank, mst, std::string("hs_merge_first_table_dev: ") + hs_last_error(h));
    *n_total = kept;
    if (kept > cap) return rfail(c, rank, HS_ERR_CAPACITY, "hit buffers too small; see *n_total");
    if (kept) {
      if (!hit_q || !hit_id || !hit_dist) return HS_ERR_INVALID;
      HSD_HIP(c, rank, hipMemcpy(hit_q, me.all_q.p, kept * 4, hipMemcpyDeviceToHost));
      HSD_HIP(c, rank, hipMemcpy(hit_id, me.all_id.p, kept * 4, hipMemcpyDeviceToHost));
      if (hit_table) HSD_HIP(c, rank, hipMemcpy(hit_table, me.all_table.p, kept * 4, hipMemcpyDeviceToHost));
      HSD_HIP(c, rank, hipMemcpy(hit_dist, me.all_dist.p, kept * 8, hipMemcpyDeviceToHost));
    }
    return HS_OK;
  }
  if (loop) {
    // ranks with devices over the host-memory transport (two ranks may share a GPU: the protocol of
    // hs_motif_both_points --gpus n on a box with fewer GPUs): this rank's hits to the host, the
    // exchange writes straight into the caller's arrays
    me.h_q.resize(n_local);
    me.h_id.resize(n_local);
    me.h_table.resize(n_local);
    me.h_dist.resize(n_local);
    if (n_local) {
      hip(hipMemcpy(me.h_q.data(), me.io_q.p, n_local * 4, hipMemcpyDeviceToHost), "hits: hipMemcpy");
      hip(hipMemcpy(me.h_id.data(), me.io_id.p, n_local * 4, hipMemcpyDeviceToHost), "hits: hipMemcpy");
      hip(hipMemcpy(me.h_table.data(), me.io_table.p, n_local * 4, hipMemcpyDeviceToHost), "hits: hipMemcpy");
      hip(hipMemcpy(me.h_dist.data(), me.io_dist.p, n_local * 8, hipMemcpyDeviceToHost), "hits: hipMemcpy");
    }
    return allgather_impl(c, rank, lst, me.h_q.data(), me.h_id.data(), hit_table ? me.h_table.data() : nullptr,
                          me.h_dist.data(), lst == HS_OK ? n_local : 0, q_offset, hit_q, hit_id, hit_table, hit_dist,
                          cap, n_total);
  }
  uint64_t acap = std::max<uint64_t>(me.all_q.cap / 4, std::max<uint64_t>(1024, cap));
  for (;;) {
    if (lst == HS_OK)
      (void)(hip(me.all_q.reserve(acap * 4), "all hits: hipMalloc") && hip(me.all_id.reserve(acap * 4), "all hits: hipMalloc") &&
             hip(me.all_table.reserve(acap * 4), "all hits: hipMalloc") && hip(me.all_dist.reserve(acap * 8), "all hits: hipMalloc"));
    if (lst != HS_OK) n_local = 0;
    st = allgather_impl(c, rank, lst, static_cast<const uint32_t*>(me.io_q.p), static_cast<const uint32_t*>(me.io_id.p),
                        static_cast<const uint32_t*>(me.io_table.p), static_cast<const double*>(me.io_dist.p),
                        n_local, q_offset, static_cast<uint32_t*>(me.all_q.p),
                        static_cast<uint32_t*>(me.all_id.p), static_cast<uint32_t*>(me.all_table.p),
                        static_cast<double*>(me.all_dist.p), acap, n_total);
    if (st == HS_ERR_CAPACITY) {  // every rank sees the same total and repeats the exchange
      acap = *n_total + 1024;
      continue;
    }
    break;
  }
  if (st != HS_OK) return st;
  if (merge) {  // the gathered tuples, merged in place on this rank's GPU
    uint64_t kept = 0;
    const hs_status mst = hs_merge_first_table_dev(h, static_cast<uint32_t*>(me.all_q.p), static_cast<uint32_t*>(me.all_id.p),
                                                   static_cast<uint32_t*>(me.all_table.p), static_cast<double*>(me.all_dist.p),
                                                   *n_total, &kept);
    if (mst != HS_OK) return rfail(c, rank, mst, std::string("hs_merge_first_table_dev: ") + hs_last_error(h));
    *n_total = kept;
  }
  const uint64_t nt = *n_total;
  if (nt > cap) return rfail(c, rank, HS_ERR_CAPACITY, "hit buffers too small; see *n_total");
  if (nt) {
    if (!hit_q || !hit_id || !hit_dist) return HS_ERR_INVALID;
    HSD_HIP(c, rank, hipMemcpy(hit_q, me.all_q.p, nt * 4, hipMemcpyDeviceToHost));
    HSD_HIP(c, rank, hipMemcpy(hit_id, me.all_id.p, nt * 4, hipMemcpyDeviceToHost));
    if (hit_table) HSD_HIP(c, rank, hipMemcpy(hit_table, me.all_table.p, nt * 4, hipMemcpyDeviceToHost));
    HSD_HIP(c, rank, hipMemcpy(hit_dist, me.all_dist.p, nt * 8, hipMemcpyDeviceToHost));
  }
  return HS_OK;
}

}  // namespace

extern "C" {

hs_status hs_comm_query(hs_comm* c, uint32_t rank, hs_handle* h, const double* centers, uint64_t nq_local,
                        uint32_t q_offset, double R, uint32_t* hit_q, uint32_t* hit_id,
                        uint32_t* hit_table, double* hit_dist, uint64_t cap, uint64_t* n_total) {
  return comm_query_impl(c, rank, h, centers, nullptr, nq_local, q_offset, R, hit_q, hit_id, hit_table, hit_dist,
                         cap, n_total);
}

hs_status hs_comm_query_tables(hs_comm* c, uint32_t rank, hs_handle* h, const uint32_t* tables, uint32_t n_tables,
                               const double* centers, const uint8_t* qcodes, uint64_t nq, double R, uint32_t* hit_q,
                               uint32_t* hit_id, uint32_t* hit_table, double* hit_dist, uint64_t cap,
                               uint64_t* n_total) {
  static const uint32_t none[1] = {0u};
  // (a rank that cannot name its tables still takes part, as a failed rank: an empty map fails the checks)
  const bool args_ok = tables && n_tables && (centers || qcodes || !nq) && !(centers && qcodes);
  return comm_query_impl(c, rank, args_ok ? h : nullptr, centers, qcodes, nq, 0u, R, hit_q, hit_id, hit_table,
                         hit_dist, cap, n_total, tables ? tables : none, tables ? n_tables : 0u);
}

hs_status hs_comm_query_buckets(hs_comm* c, uint32_t rank, hs_handle* h, const double* centers, const uint8_t* qcodes,
                                uint64_t nq, double R, uint32_t* hit_q, uint32_t* hit_id, uint32_t* hit_table,
                                double* hit_dist, uint64_t cap, uint64_t* n_total) {
  // (a bad argument list still joins the exchange, as a failed rank)
  const bool args_ok = (centers || qcodes || !nq) && !(centers && qcodes);
  return comm_query_impl(c, rank, args_ok ? h : nullptr, centers, qcodes, nq, 0u, R, hit_q, hit_id, hit_table,
                         hit_dist, cap, n_total, nullptr, 0u, true);
}

void hs_assign_tables(const double* cost, uint32_t L, uint32_t world, uint32_t* owner) {
  // longest processing time first: tables by falling cost (ties: lower number first), each to the rank with
  // the least cost so far (ties: lower rank first) -- deterministic, so every rank computes the same map
  if (!owner || !L || !world) return;
  std::vector<uint32_t> order(L);
  for (uint32_t l = 0; l < L; ++l) order[l] = l;
  if (cost)
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return cost[x] > cost[y]; });
  std::vector<double> load(world, 0.0);
  std::vector<uint32_t> count(world, 0);
  for (uint32_t i = 0; i < L; ++i) {
    const uint32_t l = order[i];
    uint32_t best = 0;
    for (uint32_t r = 1; r < world; ++r)
      if (load[r] < load[best] || (load[r] == load[best] && count[r] < count[best])) best = r;
    owner[l] = best;
    load[best] += cost ? cost[l] : 1.0;
    ++count[best];
  }
}

hs_status hs_comm_query_codes(hs_comm* c, uint32_t rank, hs_handle* h, const uint8_t* qcodes, uint64_t nq_local,
                              uint32_t q_offset, double R, uint32_t* hit_q, uint32_t* hit_id,
                              uint32_t* hit_table, double* hit_dist, uint64_t cap, uint64_t* n_total) {
  if (nq_local && !qcodes && rank_ok(c, rank) && n_total)  // (takes part, as a failed rank)
    return comm_query_impl(c, rank, nullptr, nullptr, nullptr, nq_local, q_offset, R, hit_q, hit_id, hit_table,
                           hit_dist, cap, n_total);
  return comm_query_impl(c, rank, h, nullptr, qcodes, nq_local, q_offset, R, hit_q, hit_id, hit_table, hit_dist,
                         cap, n_total);
}

}  // extern "C"
