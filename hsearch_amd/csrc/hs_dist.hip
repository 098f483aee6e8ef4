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
  std::vector<uint64_t> counts, offsets, caps;  // published by each rank for the current call
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
  c->counts.assign(world, 0);
  c->offsets.assign(world, 0);
  c->caps.assign(world, 0);
  if (kind == HS_COMM_LOOPBACK) {
    *out = c;
    return HS_OK;
  }
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
    delete c;
    return say(err, err_cap, HS_ERR_NO_DEVICE, "no usable gfx950 device");
  }
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
  c->counts.assign(world, 0);
  c->offsets.assign(world, 0);
  c->caps.assign(world, 0);
  c->ranks[0].device = device;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  if (hipSetDevice(device) != hipSuccess ||
      hipStreamCreateWithFlags(&c->ranks[0].stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return say(err, err_cap, HS_ERR_HIP, "cannot create a stream on the rank's device");
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
    if (c->kind == HS_COMM_RCCL_LOCAL) {
      (void)hipSetDevice(r.device);
      if (r.stream) (void)hipStreamSynchronize(r.stream);
      DevBuf* bufs[] = {&r.send, &r.recv, &r.counts, &r.io_centers, &r.io_q, &r.io_id, &r.io_table, &r.io_dist,
                        &r.all_q, &r.all_id, &r.all_table, &r.all_dist};
      for (DevBuf* b : bufs) b->release();
      if (r.comm) (void)rccl()->CommDestroy(r.comm);
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

hs_status hs_allgather_hits(hs_comm* c, uint32_t rank, const uint32_t* q, const uint32_t* id,
                            const uint32_t* table, const double* dist, uint64_t n_local,
                            uint32_t q_offset, uint32_t* out_q, uint32_t* out_id, uint32_t* out_table,
                            double* out_dist, uint64_t cap, uint64_t* n_total) {
  if (!rank_ok(c, rank) || !n_total) return HS_ERR_INVALID;
  if (n_local && (!q || !id || !dist)) return HS_ERR_INVALID;
  if (n_local >= (1ull << 31)) return rfail(c, rank, HS_ERR_INVALID, "more than 2^31 - 1 hits on one rank");
  RankState& me = c->rs(rank);
  const uint32_t world = c->world;
  const bool loop = c->kind == HS_COMM_LOOPBACK;
  Rccl* R = loop ? nullptr : rccl();
  if (!loop) HSD_HIP(c, rank, hipSetDevice(me.device));
  // ---- phase 1: counts and query offsets of all ranks
  if (c->per_process && world > 1) {
    HSD_HIP(c, rank, me.counts.reserve((size_t)(world + 1) * 24));
    uint64_t mine[3] = {n_local, q_offset, cap};
    char* d_mine = static_cast<char*>(me.counts.p) + (size_t)world * 24;
    HSD_HIP(c, rank, hipMemcpyAsync(d_mine, mine, 24, hipMemcpyHostToDevice, me.stream));
    HSD_NCCL(c, rank, R->AllGather(d_mine, me.counts.p, 24, ncclChar, me.comm, me.stream));
    std::vector<uint64_t> all((size_t)world * 3);
    HSD_HIP(c, rank, hipMemcpyAsync(all.data(), me.counts.p, (size_t)world * 24, hipMemcpyDeviceToHost, me.stream));
    HSD_HIP(c, rank, hipStreamSynchronize(me.stream));
    for (uint32_t r = 0; r < world; ++r) {
      c->counts[r] = all[3 * r];
      c->offsets[r] = all[3 * r + 1];
      c->caps[r] = all[3 * r + 2];
    }
  } else {
    c->counts[rank] = n_local;
    c->offsets[rank] = q_offset;
    c->caps[rank] = cap;
    c->barrier();  // everyone has published
  }
  uint64_t m = 0, total = 0;
  for (uint32_t r = 0; r < world; ++r) {
    m = std::max(m, c->counts[r]);
    total += c->counts[r];
    cap = std::min(cap, c->caps[r]);  // one decision for all ranks: the smallest capacity offered
  }
  std::vector<uint64_t> cnt(c->counts);  // private copy: the shared one is reused by the next call
  m = pad_even(m);
  *n_total = total;
  if (total > cap) {
    c->barrier();  // nobody overwrites counts before everyone has read them
    return rfail(c, rank, HS_ERR_CAPACITY, "hit buffers too small; see *n_total");
  }
  if (total && (!out_q || !out_id || !out_dist)) {
    c->barrier();
    return HS_ERR_INVALID;
  }
  const size_t rb = rec_bytes(m);
  if (loop) {
    // ---- phase 2 (host memory): publish the record, rendezvous, copy every rank's
    me.host_send.resize(std::max<size_t>(rb, 8));
    host_pack(q, id, table, dist, n_local, m, q_offset, me.host_send.data());
    c->barrier();  // all records are in place (and all counts read)
    uint64_t off = 0;
    for (uint32_t r = 0; r < world; ++r) {
      host_unpack(c->ranks[r].host_send.data(), m, cnt[r], off, out_q, out_id, out_table, out_dist);
      off += cnt[r];
    }
    c->barrier();  // nobody repacks while a neighbour still reads
    return HS_OK;
  }
  // ---- phase 2 (RCCL): pack, ONE all-gather, unpack
  if (!c->per_process) c->barrier();  // all counts read before anyone's next call republishes
  if (!m) return HS_OK;
  HSD_HIP(c, rank, me.send.reserve(rb));
  HSD_HIP(c, rank, me.recv.reserve(rb * world));
  const unsigned pb = (unsigned)((m + 255) / 256);
  hs_pack_hits_kernel<<<pb, 256, 0, me.stream>>>(q, id, table, dist, n_local, m, q_offset,
                                                 static_cast<char*>(me.send.p));
  HSD_HIP(c, rank, hipGetLastError());
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

hs_status hs_comm_query(hs_comm* c, uint32_t rank, hs_handle* h, const double* centers, uint64_t nq_local,
                        uint32_t q_offset, double R, uint32_t* hit_q, uint32_t* hit_id,
                        uint32_t* hit_table, double* hit_dist, uint64_t cap, uint64_t* n_total) {
  if (!rank_ok(c, rank) || !h || !n_total || (nq_local && !centers)) return HS_ERR_INVALID;
  if (c->kind == HS_COMM_LOOPBACK) return rfail(c, rank, HS_ERR_INVALID, "hs_comm_query needs a GPU communicator");
  RankState& me = c->rs(rank);
  HSD_HIP(c, rank, hipSetDevice(me.device));
  // this rank's block on its GPU, then the search with HBM-resident outputs
  hs_params prm;
  if (hs_get_params(h, &prm) != HS_OK) return HS_ERR_INVALID;
  if (prm.device != me.device) return rfail(c, rank, HS_ERR_INVALID, "the handle is bound to another device than the rank");
  const uint64_t d = 8ull * prm.k;
  const size_t cbytes = (size_t)nq_local * d * 8;
  HSD_HIP(c, rank, me.io_centers.reserve(std::max<size_t>(16, cbytes)));
  if (cbytes) HSD_HIP(c, rank, hipMemcpy(me.io_centers.p, centers, cbytes, hipMemcpyHostToDevice));
  uint64_t lcap = std::max<uint64_t>(me.io_q.cap / 4, std::max<uint64_t>(1024, 16 * nq_local)), n_local = 0;
  hs_status st;
  for (;;) {
    HSD_HIP(c, rank, me.io_q.reserve(lcap * 4));
    HSD_HIP(c, rank, me.io_id.reserve(lcap * 4));
    HSD_HIP(c, rank, me.io_table.reserve(lcap * 4));
    HSD_HIP(c, rank, me.io_dist.reserve(lcap * 8));
    st = hs_query_dev(h, static_cast<const double*>(me.io_centers.p), nq_local, R,
                      static_cast<uint32_t*>(me.io_q.p), static_cast<uint32_t*>(me.io_id.p),
                      static_cast<uint32_t*>(me.io_table.p), static_cast<double*>(me.io_dist.p), lcap,
                      &n_local, nullptr);
    if (st == HS_ERR_CAPACITY) {
      lcap = n_local + n_local / 8 + 1024;
      continue;
    }
    break;
  }
  // a failed rank still takes part in the exchange (with no hits), so the others do not hang
  const hs_status qst = st;
  if (qst != HS_OK) {
    me.err = std::string("hs_query_dev: ") + hs_last_error(h);
    n_local = 0;
  }
  uint64_t acap = std::max<uint64_t>(me.all_q.cap / 4, std::max<uint64_t>(1024, cap));
  for (;;) {
    HSD_HIP(c, rank, me.all_q.reserve(acap * 4));
    HSD_HIP(c, rank, me.all_id.reserve(acap * 4));
    HSD_HIP(c, rank, me.all_table.reserve(acap * 4));
    HSD_HIP(c, rank, me.all_dist.reserve(acap * 8));
    st = hs_allgather_hits(c, rank, static_cast<const uint32_t*>(me.io_q.p), static_cast<const uint32_t*>(me.io_id.p),
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
  if (qst != HS_OK) return qst;
  if (st != HS_OK) return st;
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

}  // extern "C"
