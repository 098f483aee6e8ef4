// hs_capi.hip -- the C ABI of include/hsearch.h on top of the gfx950 kernels (hs_kernels.hip) and
// the device primitives (hs_prims.hip).  One handle = one GPU, one stream, one index.
//
// HBM layout of an index (N k-mers of k residues, L tables, PW = ceil(k/25) 16-byte words):
//   codes       [N][k]      u8   original order (exact fp64 re-evaluation of survivors)
//   packed_all  [N][PW]     u128 5-bit residues, original order (brute force scans this)
//   per table l:
//     ids       [N]         u32  DB ids grouped by bucket, ascending inside a bucket
//     packed    [N][PW]     u128 the same k-mers in bucket order -> a bucket is ONE contiguous,
//                                coalesced stream for the verify kernel (no gather at query time)
//     dir_key   [nb]        u64  sorted fingerprints of the distinct HashKey strings
//     dir_start [nb+1]      u32  bucket boundaries
//     dir_tuple [nb][K]     i32  bucket ints of each bucket (exact string check at probe time)
#include <math.h>
#include <random>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/hs_tables.h"
#include "hs_internal.h"

namespace {

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
  template <class T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

// host staging kept across calls (pageable: page-locking 40 MB took longer than a whole
// Clustering() of 10^6 k-mers; what is saved is the fresh allocation + page faults per call)
struct HostBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    free(p);
    cap = 0;
    p = malloc(bytes);
    if (!p) return hipErrorOutOfMemory;
    memset(p, 0, bytes);  // touch the pages here: a device -> host copy into untouched pages crawls
    cap = bytes;
    return hipSuccess;
  }
  void release() {
    free(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

enum { EV_COUNT = 12 };
enum { EV_FORK = 0, EV_JOIN = 1, EVX_COUNT = 2 };

// Switches of a handle (hs_set_option, include/hsearch.h; read_knobs for the few that come from the
// environment).  None of them changes a result: each forces one of several equivalent paths (the tests run
// both and compare), sizes a batch, or prints a diagnostic.  Fault injection (HS_TEST_SPLIT_ABOVE) exists only
// in the test build of the library (-DHS_TEST_HOOKS: libhsearch_amd_hooks.so), never in libhsearch_amd.so.
struct Knobs {
  bool build_serial = false;       // HS_OPT_BUILD_SERIAL: no hash / sort overlap in the build (measurement)
  int join_xcd_run = -1;           // HS_OPT_JOIN_XCD_RUN: chunks per XCD-local run of join items (0 off, -1 auto)
  bool no_probe_records = false;   // HS_OPT_PROBE_RECORDS = 0: the probe reads the directory arrays, not the records
  uint32_t join_chunk = 0;         // HS_OPT_JOIN_CHUNK: items per counter access of hs_join8x_kernel (0: by itself)
  bool build_debug = false;        // HS_BUILD_DEBUG: say when a table is sorted a second time
  bool cluster_timing = false;     // HS_CLUSTER_TIMING: phase times of hs_self_join_range on stderr
  bool debug_refine = false;       // HS_DEBUG_REFINE: survivor counts per batch on stderr
  bool force_wide = false;         // HS_OPT_WIDE_ROWS = 1: 8-column rows whatever the radius (k <= 25)
  bool no_wide_by_radius = false;  // HS_OPT_WIDE_ROWS >= 2: never choose 8-column rows by radius
  bool no_refine8 = false;         // HS_OPT_REFINE8 = 0: no 8-column refinement of the join's survivors
  bool no_self_codes = false;      // HS_OPT_SELF_CODES = 0: self-join from embedded centres, not from codes
  bool sort_hits = false;          // HS_OPT_SORT_HITS: order hits by the radix sort, not per query
  bool sync_items = false;         // HS_OPT_SYNC_ITEMS: read the join's item count back before launching it
  bool no_join_r = false;          // HS_OPT_JOIN_RESIDENT = 1: every segment through the query-streaming join kernel
  bool no_recognise = false;       // HS_OPT_RECOGNISE_KMERS = 0: centres that are k-mers are not looked for (run_query)
  bool force_join_r = false;       // HS_OPT_JOIN_RESIDENT = 2: the query-resident kernel for its class whatever its share
  bool build_sort = false;         // HS_OPT_BUILD_GROUPING = 1: group a table's k-mers by sorting (fingerprint, id) pairs
                                   // (rocPRIM; rounds 1-2) instead of hs_group.hip's table + rank sort
  int seg_mode = 0;                // HS_OPT_SEG_MODE: 1 sparse / 2 dense; 0 = by the bucket : probe ratio
  int sort_from_bit = 16;          // HS_OPT_SORT_FROM_BIT: lowest fingerprint bit the build's sort looks at
  uint32_t query_batch = 0;        // HS_OPT_QUERY_BATCH: queries per batch (0: by L and the free HBM)
#ifdef HS_TEST_HOOKS
  uint32_t test_split_above = 0;   // HS_TEST_SPLIT_ABOVE: batches above this size report a survivor overflow
  bool test_group_fallback = false;  // HS_TEST_GROUP_FALLBACK: the build's fingerprint table reports itself full
#endif
};

}  // namespace

struct hs_handle {
  hs_params p;
  int d = 0, LK = 0, PW = 0, alphabet = HS_ALPHABET;
  int n_cu = 256;
  hipStream_t stream = nullptr;
  hipEvent_t ev[EV_COUNT];
  bool ev_ok = false;
  // side stream: the streaming filter (and its per-query tables) runs beside the bucket join
  hipStream_t stream2 = nullptr;
  hipEvent_t evx[EVX_COUNT];
  bool evx_ok = false;
  DevBuf a, b, coords;
  DevBuf aT;  // the planes transposed, [d][L*K]: what the hash kernels read
  // index
  bool built = false;
  uint64_t n = 0;
  uint32_t key_seed = 0;
  DevBuf codes, packed_all;
  DevBuf t_dirkey[HS_MAX_L], t_dirstart[HS_MAX_L], t_dirtuple[HS_MAX_L], t_ids[HS_MAX_L];
  DevBuf t_dirjump;  // the jump tables of all directories, one allocation (a table's at its own offset)
  DevBuf t_dirrec;   // the directory records of all tables (hs_table_dev::dir_rec), 64 bytes per bucket
  DevBuf part_work;  // bucket partition: flags, positions, compacted (bucket, probe) pairs of a batch
  DevBuf t_giant;    // the fingerprints of every table's giant buckets (hs_table_dev::giant_key), ascending
  uint32_t bucket_part = 0, bucket_parts = 1;  // hs_set_bucket_partition
  // bucket-ordered packed copies of all tables in ONE allocation ([L][n][PW]), and the int8 join's
  // per-entry records ([L][n], k <= 50 only) at the same entry numbers
  DevBuf t_packed, t_rec8;
  // the records once more in four bytes per entry ([L][n]; k <= 25 with 4-column rows: what the
  // query-resident join kernel reads instead of t_rec8 -- it is bound by the bytes it moves per member)
  DevBuf t_rho;
  DevBuf hit_kv;     // query_batch: the hits bucketed by query, (key, value) side by side
  DevBuf hit_rank;   // query_batch: a hit's number among its query's hits (ordering without a sort)
  DevBuf qpacked;    // query_batch: queries given as k-mers, packed like the members (exact pass)
  DevBuf rec_codes;  // run_query: the residue codes of centres that turned out to be k-mers (+ the counter)
  // member records of the wide rows for k = 21..25 (built when a call's radius first asks for them)
  DevBuf t_rec8w;
  bool rec8w_ready = false;
  double pair4_mean = 0.0, pair4_var = 0.0;  // 4-column squared distance of two random residues
  DevBuf t_pos;  // [L][n] sorted position of every DB id in every table (first-seen dedupe)
  DevBuf dir_base;       // [L + 1] first global bucket number of every table; [L] = nb_total
  uint32_t nb_total = 0;  // buckets of all tables
  DevBuf bucket_work;    // per-batch: counts + 3 work arrays over the nb_total + 2 bucket slots
  hs_tables_dev tabs;
  hs_index_info info;
  // query workspace (grown on demand, reused across calls)
  DevBuf qints, qstart, qcount, nslices, slice_off, tq, prov, hit_key, hit_val, hit_key2, hit_val2,
      counters, temp, io_centers, io_q, io_id, io_table, io_dist, io_cand, io_codes, io_misc;
  DevBuf qcodes_buf, qembed;  // hs_query_codes: a batch's checked copy of the query codes; their embedding
                              // when no from-codes path applies
  HostBuf sj_host;  // hs_self_join_range: hits of one chunk on their way to the edge lists
  // bucket-join workspace
  // hs_index_build_subset: the caller's whole code array, kept on the device across calls
  DevBuf all_codes, subset_ids;
  const uint8_t* all_codes_key = nullptr;
  uint64_t all_codes_n = 0;
  DevBuf bs_ints2[2], bs_keys2[2], bs_iota2[2], bs_keys_sorted, bs_rle_unique, bs_rle_counts, bs_small,
      bs_sort_temp, bs_slow_q;  // index-build scratch (build_tables)
  DevBuf bs_fptab, bs_blk, bs_dk, bs_hist, bs_rank;  // ... of the table + rank-sort grouping (hs_group.hip)
  // hs_index_shard_*: the build with the hashing spread over ranks (this rank's block of the k-mers)
  bool shard_open = false;
  uint32_t shard_lo = 0, shard_cnt = 0, shard_seed = 0, shard_nb = 0;
  int shard_table = -1;
  DevBuf seg_of;    // query_batch: the segment of every sorted probe position
  DevBuf seg_res;   // cut_items: flags + scan of the segments that go to the query-resident join kernel
  DevBuf c16s, item_desc, probe_slow, jtab8, qhits;  // qhits: per-query hit counts, offsets, fill
  DevBuf c8b, prov2;  // survivor refinement: second int8 row per query, the refined survivor list
  bool join8_tables_ok = false;  // int8 can carry the coordinate table
  double join8_scale = 0.0, join8_scale_w = 0.0;  // quantisation scales: 4-column rows, wide rows
  bool wide8_ok = false;         // the 8-column table is usable (wide rows on demand for k = 21..25)
  uint32_t* pin_cnt = nullptr;   // 64 pinned words: where a batch's counters land (three small device ->
                                 // host copies into PAGEABLE memory cost ~ 50 us of host staging per batch)
  double pairs_per_item = 0.0;   // average of the previous batch's join work items (0: none yet)
  bool order_failed = false;     // the last batch that ordered its hits itself had to fall back to the sort
  double order_failed_R = 0.0;   // ... at this radius
  Knobs knobs;
  bool wide8 = false;            // short k-mers: int8 rows over all 8 coordinate columns (hs_join8.hip)
  // segment routing thresholds (HS_JOIN_MIN_Q / _M): segments with fewer probing queries or members
  // go to the per-pair filters instead of the join.  1 / 1 = everything through the join: its
  // persistent waves leave no room for a kernel beside it, and the per-pair filter run before it
  // cost 0.3 ms at C2 (a chain of dependent loads per probe) against 0.06 ms of extra join time
  uint32_t join_min_q = 1, join_min_m = 1;
  // work items of the last joined batch x 1.25: with it the next batch sizes its descriptor array
  // without asking the device (the kernels clamp to the real count; an overflow repeats the batch)
  uint32_t item_cap_hint = 0;
  // share of the last joined batch's work items that lay in segments with few probing queries (the
  // query-resident kernel's class); < 0: unknown.  That kernel pays when the class is the bulk of the items
  // (configs[2]'s shape: 90 %); where it is a minority (configs[1]: the extra launch costs more than the
  // class's items cost in the streaming kernel) the next batch runs everything through the streaming kernel
  double resident_share = -1.0;
  uint32_t resident_age = 0;     // joined batches since it was measured (measured again every 64)
  uint32_t resident_nq = 0;      // ... on a batch of this many queries (a batch half / twice that size measures anew)
  int join_blocks_per_cu = 2;                // resident workgroups of hs_join_kernel per CU
  DevBuf jtab, c16, seg_keys, seg_keys_sorted, seg_vals, sorted_ql, seg_key, seg_cnt, seg_qoff,
      seg_items, item_off, seg_n;
  // projection on the matrix cores (hs_proj.hip): quantised planes in both tilings, per-function
  // constants, the quantised coordinate table, flag lists (0/1: the two tables in flight during a
  // build, 2: queries and the hash entry points) and their counters {reserved, real} x 3
  int proj_S = 0;               // k-steps the MFMA pass is compiled for (0: k too long)
  bool proj_usable = false;     // S > 0 and the buffers below are filled for the current planes
  bool proj_auto = false;       // ... and the error bound is narrow enough for the auto mode
  int hash_mode = 0;            // 0 auto, 1 exact fp64 kernel, 2 MFMA pass wherever usable
  double proj_eps_scale = 1.0;
  double proj_est = 0.0;        // typical half-width of the bound, in bucket units
  DevBuf proj_aq_all, proj_aq_tab, proj_fn, proj_tab, proj_stats, proj_flags[3], proj_cnt, proj_xq, proj_xmeta;
  bool sqrt_test = false;       // hit test sqrt(d2) <= R (hclust2.cpp:119-120) instead of d2 <= R*R
  uint32_t self_first = HS_NO_SELF;  // self-join: DB id of query 0 of the current run_query
  bool join_tables_ok = false;  // fp16 can carry the coordinate table
  int verify_mode = 0;          // 0 auto, 1 streaming kernel, 2 bucket join
  std::string err;
  hs_profile prof;
};

namespace {

hs_status fail(hs_handle* h, hs_status st, const std::string& msg) {
  if (h) h->err = msg;
  return st;
}

#define HS_HIP(h, expr)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(h, e_ == hipErrorOutOfMemory ? HS_ERR_NOMEM : HS_ERR_HIP,                  \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                        \
  } while (0)
#define HS_CHECK(expr)                \
  do {                                \
    hs_status st_ = (expr);           \
    if (st_ != HS_OK) return st_;     \
  } while (0)

float ev_ms(hs_handle* h, int i0, int i1) {
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, h->ev[i0], h->ev[i1]) != hipSuccess) return 0.f;
  return ms;
}

// smallest float >= x (x >= 0), then one more ulp: the fp32 filter bound must never undercut.
float filter_bound(double r2) {
  double hi = r2 * (1.0 + 1e-5) + 1e-30;
  float f = (float)hi;
  if ((double)f < hi) f = nextafterf(f, INFINITY);
  return nextafterf(f, INFINITY);
}

// The index is about to change (or go): nothing learnt from batches against the old one may size or
// steer batches against the new one (a stale capacity hint made the first batches after a rebuild to
// another shape run their join twice: once with the stale capacity, then again the slow way).
void drop_index(hs_handle* h) {
  h->built = false;
  h->rec8w_ready = false;
  h->order_failed = false;
  h->order_failed_R = 0.0;
  h->item_cap_hint = 0;
  h->pairs_per_item = 0.0;
  h->resident_share = -1.0;
  h->resident_age = 0;
  h->resident_nq = 0;
}

hs_status ensure_device(hs_handle* h) {
  HS_HIP(h, hipSetDevice(h->p.device));
  return HS_OK;
}


// The device code of the library's kernels -- rocPRIM's sort / scan / run-length kernels in
// particular -- is loaded lazily, at first launch: measured 57 ms of host time inside the first
// table's sort of the first index build of a process (rocprofv3 timeline).  One launch of each
// on a few elements at handle creation moves that out of hs_index_build and out of the first query.
hs_status warm_up_device_code(hs_handle* h) {
  static bool done = false;  // per process
  if (done) return HS_OK;
  const size_t n = 64;
  DevBuf k0, k1, v0, v1, cnt, tmp;
  struct G {
    DevBuf* b[6];
    ~G() { for (DevBuf* x : b) x->release(); }
  } g = {{&k0, &k1, &v0, &v1, &cnt, &tmp}};
  HS_HIP(h, k0.reserve(n * 8));
  HS_HIP(h, k1.reserve(n * 8));
  HS_HIP(h, v0.reserve(n * 8));
  HS_HIP(h, v1.reserve(n * 8));
  HS_HIP(h, cnt.reserve((n + 2) * 4));
  const size_t tb = std::max(std::max(hs_sort_pairs_u64_u32_temp(n), hs_sort_pairs_u64_u64_temp(n)),
                             std::max(hs_rle_u64_temp(n), hs_scan_u32_temp(n))) + 256;
  HS_HIP(h, tmp.reserve(tb));
  HS_HIP(h, hipMemsetAsync(k0.p, 0, n * 8, h->stream));
  HS_HIP(h, hipMemsetAsync(v0.p, 0, n * 8, h->stream));
  HS_HIP(h, hs_sort_pairs_u64_u32(tmp.p, tmp.cap, k0.as<uint64_t>(), k1.as<uint64_t>(), v0.as<uint32_t>(),
                                  v1.as<uint32_t>(), n, 0, 64, h->stream));
  HS_HIP(h, hs_sort_pairs_u64_u64(tmp.p, tmp.cap, k0.as<uint64_t>(), k1.as<uint64_t>(), v0.as<uint64_t>(),
                                  v1.as<uint64_t>(), n, 64, h->stream));
  HS_HIP(h, hs_rle_u64(tmp.p, tmp.cap, k1.as<uint64_t>(), k0.as<uint64_t>(), v0.as<uint32_t>(),
                       cnt.as<uint32_t>(), n, h->stream));
  HS_HIP(h, hs_exclusive_scan_u32(tmp.p, tmp.cap, v0.as<uint32_t>(), v1.as<uint32_t>(), n, h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  done = true;
  return HS_OK;
}

// (Re)quantise the handle's planes -- and, the first time, its coordinate table -- for the MFMA
// projection, and decide whether the auto mode uses it: the bound's typical half-width, in bucket
// units, is est = (da k max|row|_1 + dx max|a^|_1) / W; about 2 est of all values are recomputed.
hs_status setup_projection(hs_handle* h, bool table_too) {
  h->proj_usable = false;
  h->proj_auto = false;
  h->proj_S = hs_proj_steps((int)h->p.k);
  if (!h->proj_S) return HS_OK;
  const int S = h->proj_S, LK = h->LK, L = (int)h->p.L;
  const size_t tile = (size_t)S * 2 * 64 * 16;
  const size_t all_bytes = (size_t)((LK + 31) / 32) * tile, tab_bytes = (size_t)L * tile;
  HS_HIP(h, h->proj_aq_all.reserve(all_bytes));
  HS_HIP(h, h->proj_aq_tab.reserve(tab_bytes));
  HS_HIP(h, h->proj_fn.reserve((size_t)LK * 32));
  HS_HIP(h, h->proj_tab.reserve(sizeof(hs_proj_table)));
  HS_HIP(h, h->proj_stats.reserve(64));
  HS_HIP(h, h->proj_cnt.reserve(64));
  HS_HIP(h, hipMemsetAsync(h->proj_aq_all.p, 0, all_bytes, h->stream));
  HS_HIP(h, hipMemsetAsync(h->proj_aq_tab.p, 0, tab_bytes, h->stream));
  HS_HIP(h, hipMemsetAsync(h->proj_stats.p, 0, 64, h->stream));
  if (table_too)
    HS_HIP(h, hs_launch_quant_table(h->coords.as<double>(), h->alphabet, h->proj_tab.as<hs_proj_table>(), h->stream));
  HS_HIP(h, hs_launch_quant_planes(h->a.as<double>(), h->b.as<double>(), LK, h->d, (int)h->p.K, S, h->p.W,
                                   h->proj_eps_scale, h->proj_aq_all.p, h->proj_aq_tab.p, h->proj_fn.p,
                                   h->proj_stats.as<unsigned long long>(), h->stream));
  unsigned long long stats[3] = {0, 0, 0};
  hs_proj_table tab;
  HS_HIP(h, hipMemcpyAsync(stats, h->proj_stats.p, 24, hipMemcpyDeviceToHost, h->stream));
  HS_HIP(h, hipMemcpyAsync(&tab, h->proj_tab.p, sizeof(tab), hipMemcpyDeviceToHost, h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  double da, a1;
  memcpy(&da, &stats[0], 8);
  memcpy(&a1, &stats[1], 8);
  h->proj_usable = true;
  h->proj_est = h->proj_eps_scale * (da * (double)h->p.k * tab.l1max + tab.dx * a1) / h->p.W;
  // auto: at most ~3 % of the values recomputed, every function and the table representable
  h->proj_auto = !tab.unsafe && !stats[2] && h->proj_est <= 1.0 / 64.0;
  return HS_OK;
}

inline bool use_projection(const hs_handle* h) {
  return h->proj_usable && (h->hash_mode == 2 || (h->hash_mode == 0 && h->proj_auto));
}

// Bucket ints of functions [f0, f0 + F) -- one whole table (table >= 0: f0 = table * K, F = K) or
// all of them (table < 0) -- for n points given as codes or as doubles, into out[i * out_stride + f - f0],
// on stream s: the MFMA pass + exact recomputation of the flagged values, or the exact kernel alone.
// set = which flag list / counter pair to use (two tables are in flight during a build).
hs_status hash_dispatch(hs_handle* h, const uint8_t* d_codes, const double* d_pts, uint64_t n, int table,
                        int32_t* out, int out_stride, int set, hipStream_t s) {
  const int K = (int)h->p.K, k = (int)h->p.k;
  const int f0 = table >= 0 ? table * K : 0, F = table >= 0 ? K : h->LK;
  if (!n) return HS_OK;
  if (!use_projection(h) || n >= (1ull << 31)) {
    if (d_codes)
      HS_HIP(h, hs_launch_hash_codes(d_codes, n, k, h->aT.as<double>() + f0, h->LK, h->b.as<double>() + f0, F,
                                     h->p.W, h->coords.as<double>(), out, out_stride, s));
    else
      HS_HIP(h, hs_launch_hash_points(d_pts, n, k, h->aT.as<double>() + f0, h->LK, h->b.as<double>() + f0, F,
                                      h->p.W, out, out_stride, s));
    return HS_OK;
  }
  // the flag list's counter is 32 bits and every value may be flagged (an inflated bound, a table the
  // fixed point cannot carry): at most 0xE0000000 / F points per pass
  const uint64_t n_max = 0xE0000000ull / (uint64_t)F;
  if (n > n_max) {
    for (uint64_t i0 = 0; i0 < n; i0 += n_max)
      HS_CHECK(hash_dispatch(h, d_codes ? d_codes + i0 * k : nullptr, d_pts ? d_pts + i0 * h->d : nullptr,
                             std::min(n_max, n - i0), table, out + i0 * out_stride, out_stride, set, s));
    return HS_OK;
  }
  const int S = h->proj_S;
  const size_t tile = (size_t)S * 2 * 64;  // uint4 per function tile
  const uint4* aq = table >= 0 ? h->proj_aq_tab.as<uint4>() + (size_t)table * tile : h->proj_aq_all.as<uint4>();
  const char* fn = h->proj_fn.as<char>() + (size_t)f0 * 32;
  // room for 1/32 of the values (the auto mode expects ~ 2 est <= 1/32 of them at worst) plus the
  // slack of the per-wave reservations (2 n_cu x 4 waves x 128 slots); when the list overflows the
  // fix kernel recomputes everything, which is still correct
  const uint64_t want = (uint64_t)n * (uint64_t)F / 32 + (1u << 20);
  const uint32_t cap = (uint32_t)std::min<uint64_t>(want, 1ull << 28);
  HS_HIP(h, h->proj_flags[set].reserve((size_t)cap * 8));
  uint32_t* cnt = h->proj_cnt.as<uint32_t>() + 2 * set;
  HS_HIP(h, hipMemsetAsync(cnt, 0, 8, s));
  if (!d_codes) {
    HS_HIP(h, h->proj_xq.reserve((size_t)n * S * 64));
    HS_HIP(h, h->proj_xmeta.reserve((size_t)n * 24));
    HS_HIP(h, hs_launch_quant_points(d_pts, n, k, S, h->proj_xq.p, h->proj_xmeta.as<double>(), s));
  }
  HS_HIP(h, hs_launch_proj(d_codes, h->proj_xq.p, h->proj_xmeta.as<double>(), n, k, S, aq, fn, F,
                           h->proj_tab.as<hs_proj_table>(), h->p.W, out, out_stride,
                           h->proj_flags[set].as<uint2>(), cap, cnt, h->n_cu, s));
  HS_HIP(h, hs_launch_proj_fix(d_codes, d_pts, n, k, h->aT.as<double>() + f0, h->LK, h->b.as<double>() + f0, F,
                               h->p.W, h->coords.as<double>(), out, out_stride, h->proj_flags[set].as<uint2>(),
                               cap, cnt, s));
  return HS_OK;
}

// after a synchronisation of the stream the hash ran on: add the pass's statistics to the profile
hs_status hash_account(hs_handle* h, uint64_t n, int F, int set) {
  if (!use_projection(h) || !n || n >= (1ull << 31)) return HS_OK;
  uint32_t c[2] = {0, 0};
  // (on the handle's stream, which the callers have just drained: a plain hipMemcpy goes through the
  // process's default stream, whose first use cost 6.8 ms of a first build's 30)
  HS_HIP(h, hipMemcpyAsync(c, h->proj_cnt.as<uint32_t>() + 2 * set, 8, hipMemcpyDeviceToHost, h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  h->prof.hash_values += n * (uint64_t)F;
  h->prof.hash_flagged += c[1];
  return HS_OK;
}

// The environment is read ONCE per handle, for diagnostics and defaults only: HS_BUILD_DEBUG / HS_CLUSTER_TIMING
// / HS_DEBUG_REFINE (prints), HS_VERIFY_MODE / HS_HASH_MODE (documented defaults of hs_set_verify_mode /
// hs_set_hash_mode), HS_OPTIONS = "name=value,..." (hs_set_option by name, for the A/B scripts under tools/),
// and -- test build of the library only -- the fault-injection hooks.  Every path selection is an hs_option.
const struct { const char* name; int option; } kOptionNames[] = {
    {"query_batch", HS_OPT_QUERY_BATCH}, {"seg_mode", HS_OPT_SEG_MODE}, {"join_resident", HS_OPT_JOIN_RESIDENT},
    {"recognise_kmers", HS_OPT_RECOGNISE_KMERS}, {"build_grouping", HS_OPT_BUILD_GROUPING}, {"wide_rows", HS_OPT_WIDE_ROWS},
    {"refine8", HS_OPT_REFINE8}, {"self_codes", HS_OPT_SELF_CODES},
    {"sort_hits", HS_OPT_SORT_HITS}, {"sync_items", HS_OPT_SYNC_ITEMS}, {"join_min_q", HS_OPT_JOIN_MIN_Q},
    {"join_min_m", HS_OPT_JOIN_MIN_M}, {"sort_from_bit", HS_OPT_SORT_FROM_BIT}, {"build_serial", HS_OPT_BUILD_SERIAL},
    {"join_xcd_run", HS_OPT_JOIN_XCD_RUN}, {"probe_records", HS_OPT_PROBE_RECORDS},
    {"join_chunk", HS_OPT_JOIN_CHUNK}};

void read_knobs(hs_handle* h) {
  Knobs& kn = h->knobs;
  auto on = [](const char* name) { return getenv(name) != nullptr; };
  kn.build_debug = on("HS_BUILD_DEBUG");
  kn.cluster_timing = on("HS_CLUSTER_TIMING");
  kn.debug_refine = on("HS_DEBUG_REFINE");
#ifdef HS_TEST_HOOKS
  if (const char* m = getenv("HS_TEST_SPLIT_ABOVE")) kn.test_split_above = (uint32_t)std::max(0, atoi(m));
  kn.test_group_fallback = on("HS_TEST_GROUP_FALLBACK");
#endif
  if (const char* m = getenv("HS_HASH_MODE")) {
    if (!strcmp(m, "exact")) h->hash_mode = 1;
    if (!strcmp(m, "mfma")) h->hash_mode = 2;
  }
  if (const char* m = getenv("HS_VERIFY_MODE")) {
    if (!strcmp(m, "stream")) h->verify_mode = 1;
    if (!strcmp(m, "join")) h->verify_mode = 2;
    if (!strcmp(m, "join16")) h->verify_mode = 3;
  }
  if (const char* m = getenv("HS_OPTIONS")) {
    std::string all(m);
    for (size_t at = 0; at < all.size();) {
      const size_t end = std::min(all.find(',', at), all.size());
      const std::string item = all.substr(at, end - at);
      at = end + 1;
      const size_t eq = item.find('=');
      if (eq == std::string::npos) continue;
      for (const auto& o : kOptionNames)
        if (item.substr(0, eq) == o.name) (void)hs_set_option(h, o.option, atoll(item.c_str() + eq + 1));
    }
  }
}

}  // namespace

extern "C" {

#ifdef HS_TEST_HOOKS
const char* hs_version(void) { return "hsearch_amd 0.3 (gfx950, test hooks)"; }
#else
const char* hs_version(void) { return "hsearch_amd 0.3 (gfx950)"; }
#endif
const char* hs_last_error(const hs_handle* h) { return h ? h->err.c_str() : "null handle"; }

hs_status hs_get_profile(const hs_handle* h, hs_profile* out) {
  if (!h || !out) return HS_ERR_INVALID;
  *out = h->prof;
  return HS_OK;
}

hs_status hs_get_params(const hs_handle* h, hs_params* out) {
  if (!h || !out) return HS_ERR_INVALID;
  *out = h->p;
  out->alphabet = (uint32_t)h->alphabet;
  return HS_OK;
}

uint32_t hs_key_string(const int32_t* buckets, uint32_t K, char* out, uint32_t cap) {
  if (K > HS_MAX_K) K = HS_MAX_K;
  char tmp[HS_KEY_CHARS];
  int n = hs_key_chars(buckets, (int)K, tmp);
  if (cap) {
    uint32_t m = std::min<uint32_t>((uint32_t)n, cap - 1);
    memcpy(out, tmp, m);
    out[m] = 0;
  }
  return (uint32_t)n;
}

uint64_t hs_key_fingerprint(const int32_t* buckets, uint32_t K, uint32_t seed) {
  return hs_key_of(buckets, (int)std::min<uint32_t>(K, HS_MAX_K), seed);
}

int hs_key_strings_equal(const int32_t* x, const int32_t* y, uint32_t K) {
  return hs_key_equal(x, y, (int)std::min<uint32_t>(K, HS_MAX_K)) ? 1 : 0;
}

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void hs_abort_backtrace(int) {
  void* frames[64];
  (void)!write(2, "HS ABORT\n", 9);
  const int nf = backtrace(frames, 64);
  backtrace_symbols_fd(frames, nf, 2);
  _exit(134);
}

hs_status hs_create(const hs_params* params, const double* a, const double* b, const double* coords,
                    hs_handle** out) {
  if (!out) return HS_ERR_INVALID;
  *out = nullptr;
  if (!params || !a || !b) return HS_ERR_INVALID;
  if (params->k < 1 || params->k > 75 || params->K < 1 || params->K > HS_MAX_K || params->L < 1 ||
      params->L > HS_MAX_L || !(params->W > 0.0) || !isfinite(params->W) ||
      params->alphabet > HS_ALPHABET_PAD || (params->alphabet && !coords))
    return HS_ERR_INVALID;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || params->device < 0 ||
      params->device >= n_dev)
    return HS_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, params->device) != hipSuccess) return HS_ERR_NO_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return HS_ERR_NO_DEVICE;  // CDNA4 code objects only
  hs_handle* h = new hs_handle();
  h->p = *params;
  h->d = 8 * (int)params->k;
  h->LK = (int)(params->L * params->K);
  h->PW = hs_packed_words((int)params->k);
  h->alphabet = params->alphabet ? (int)params->alphabet : HS_ALPHABET;
  h->n_cu = prop.multiProcessorCount;
  memset(&h->tabs, 0, sizeof(h->tabs));
  memset(&h->info, 0, sizeof(h->info));
  memset(&h->prof, 0, sizeof(h->prof));
  *out = h;  // returned even on failure below so the caller can read hs_last_error, then destroy
  HS_HIP(h, hipSetDevice(params->device));
  HS_HIP(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  for (int i = 0; i < EV_COUNT; ++i) HS_HIP(h, hipEventCreate(&h->ev[i]));
  h->ev_ok = true;
  {
    // lowest priority: the runtime keeps a separate pool of hardware queues per priority level, so
    // the side stream never shares a queue with the main stream (with the default 4 queues and a
    // few more streams in the process -- torch's, RCCL's -- two normal-priority streams can land on
    // the same hardware queue, and the streaming filter would then run AFTER the join instead of
    // beside it); it is also the right order of service
    int least = 0, greatest = 0;
    HS_HIP(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
    HS_HIP(h, hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, least));
  }
  for (int i = 0; i < EVX_COUNT; ++i)
    HS_HIP(h, hipEventCreateWithFlags(&h->evx[i], hipEventDisableTiming));
  h->evx_ok = true;
  const size_t na = (size_t)h->LK * h->d;
  HS_HIP(h, h->a.reserve(na * 8));
  HS_HIP(h, h->b.reserve((size_t)h->LK * 8));
  HS_HIP(h, h->coords.reserve(HS_ALPHABET_PAD * 8 * 8));
  HS_HIP(h, hipMemsetAsync(h->coords.p, 0, HS_ALPHABET_PAD * 8 * 8, h->stream));
  HS_HIP(h, hipMemcpyAsync(h->a.p, a, na * 8, hipMemcpyHostToDevice, h->stream));
  // the hash kernels read the planes dimension-major: aT[i][f], f = l * K + k
  HS_HIP(h, h->aT.reserve(na * 8));
  HS_HIP(h, hs_launch_transpose_f64(h->a.as<double>(), h->LK, h->d, h->aT.as<double>(), h->stream));
  HS_HIP(h, hipMemcpyAsync(h->b.p, b, (size_t)h->LK * 8, hipMemcpyHostToDevice, h->stream));
  HS_HIP(h, hipMemcpyAsync(h->coords.p, coords ? coords : &HS_AA_COORDS[0][0],
                           (size_t)h->alphabet * 8 * 8, hipMemcpyHostToDevice, h->stream));
  // fp16 coordinate table + row norms of the bucket-join filter
  HS_HIP(h, h->jtab.reserve(512 + 128 + 64));
  HS_HIP(h, hipMemsetAsync(h->jtab.p, 0, 512 + 128 + 64, h->stream));
  HS_HIP(h, hs_launch_jtables(h->coords.as<double>(), h->alphabet, h->jtab.p,
                              reinterpret_cast<float*>(h->jtab.as<char>() + 512),
                              reinterpret_cast<uint32_t*>(h->jtab.as<char>() + 640), h->stream));
  uint32_t unsafe = 1;
  HS_HIP(h, hipMemcpyAsync(&unsafe, h->jtab.as<char>() + 640, 4, hipMemcpyDeviceToHost, h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  h->join_tables_ok = (unsafe == 0);
  // int8 table {x^ packed, |x1|^2, L1(x^)} x 32 at bytes 0..511, scale {s, s^2/2} at 512, flag at 640
  // ... refinement table {x^ 0..3, x^ 4..7, |x|^2, L1s} x 32 at bytes 1024..1535, scale[2..3] = its s, ok
  HS_HIP(h, h->jtab8.reserve(2048));
  HS_HIP(h, hipMemsetAsync(h->jtab8.p, 0, 2048, h->stream));
  // ... the 8-column one-scale table of the wide rows (short k-mers) at bytes 1536..2047, scale[4..6]
  HS_HIP(h, hs_launch_jtables8(h->coords.as<double>(), h->alphabet, h->jtab8.p, h->jtab8.as<float>() + 128,
                               reinterpret_cast<uint32_t*>(h->jtab8.as<char>() + 640),
                               h->jtab8.as<char>() + 1024, h->jtab8.as<char>() + 1536, h->stream));
  uint32_t unsafe8 = 1;
  float scale8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  HS_HIP(h, hipMemcpyAsync(&unsafe8, h->jtab8.as<char>() + 640, 4, hipMemcpyDeviceToHost, h->stream));
  HS_HIP(h, hipMemcpyAsync(scale8, h->jtab8.as<char>() + 512, 32, hipMemcpyDeviceToHost, h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  h->join8_tables_ok = (unsafe8 == 0);
  // Short k-mers: R^2 is not far below the 4-column distance of bucket mates any more (k = 15: the
  // 4-column bound passes 4 % of random pairs), so their rows carry all 8 columns (hs_join8.hip)
  if (getenv("HS_BACKTRACE")) signal(SIGABRT, hs_abort_backtrace);
  h->wide8_ok = h->join8_tables_ok && scale8[6] > 0.f;
  h->wide8 = h->wide8_ok && (int)h->p.k <= 20;  // (hs_set_option(HS_OPT_WIDE_ROWS, 3): never)
  h->join8_scale = (double)scale8[0];
  h->join8_scale_w = (double)scale8[4];
  {  // mean and variance of the 4-column squared distance of two uniformly drawn residues (want_wide)
    const double* ct = coords ? coords : &HS_AA_COORDS[0][0];
    const int A = h->alphabet;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < A; ++i)
      for (int j = 0; j < A; ++j) {
        double d = 0.0;
        for (int c = 0; c < 4; ++c) d += (ct[i * 8 + c] - ct[j * 8 + c]) * (ct[i * 8 + c] - ct[j * 8 + c]);
        s1 += d;
        s2 += d * d;
      }
    h->pair4_mean = s1 / ((double)A * A);
    h->pair4_var = std::max(0.0, s2 / ((double)A * A) - h->pair4_mean * h->pair4_mean);
  }
  read_knobs(h);
  HS_CHECK(setup_projection(h, true));
  HS_CHECK(warm_up_device_code(h));
  return HS_OK;
}

hs_status hs_set_planes(hs_handle* h, const double* a, const double* b) {
  if (!h || !a || !b) return HS_ERR_INVALID;
  hs_status st = ensure_device(h);
  if (st) return st;
  drop_index(h);  // the tables were keyed by the old family
  const size_t na = (size_t)h->LK * h->d;
  // the previous family may still be read by work queued on the stream: order the copies after it
  HS_HIP(h, hipMemcpyAsync(h->a.p, a, na * 8, hipMemcpyHostToDevice, h->stream));
  HS_HIP(h, hs_launch_transpose_f64(h->a.as<double>(), h->LK, h->d, h->aT.as<double>(), h->stream));
  HS_HIP(h, hipMemcpyAsync(h->b.p, b, (size_t)h->LK * 8, hipMemcpyHostToDevice, h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));  // a, b are the caller's again
  return setup_projection(h, false);
}

hs_status hs_set_hash_mode(hs_handle* h, int mode, double eps_scale) {
  if (!h || mode < 0 || mode > 2 || !(eps_scale >= 1.0) || !(eps_scale < 1e12)) return HS_ERR_INVALID;
  hs_status st = ensure_device(h);
  if (st) return st;
  h->hash_mode = mode;
  if (eps_scale != h->proj_eps_scale) {
    h->proj_eps_scale = eps_scale;
    HS_HIP(h, hipStreamSynchronize(h->stream));
    if (h->stream2) HS_HIP(h, hipStreamSynchronize(h->stream2));
    return setup_projection(h, false);
  }
  return HS_OK;
}

hs_status hs_set_verify_mode(hs_handle* h, int mode) {
  if (!h || mode < 0 || mode > 3) return HS_ERR_INVALID;
  h->verify_mode = mode;
  return HS_OK;
}

hs_status hs_set_bucket_partition(hs_handle* h, uint32_t part, uint32_t n_parts) {
  if (!h) return HS_ERR_INVALID;
  if (!n_parts || part >= n_parts || n_parts > 65536u)
    return fail(h, HS_ERR_INVALID, "hs_set_bucket_partition: part < n_parts, 1 <= n_parts <= 65536");
  h->bucket_part = part;
  h->bucket_parts = n_parts;
  return HS_OK;
}

hs_status hs_set_option(hs_handle* h, int option, int64_t value) {
  if (!h) return HS_ERR_INVALID;
  Knobs& kn = h->knobs;
  auto flag = [&](bool* dst, bool invert) -> hs_status {
    if (value != 0 && value != 1) return fail(h, HS_ERR_INVALID, "hs_set_option: the option takes 0 or 1");
    *dst = invert ? value == 0 : value == 1;
    return HS_OK;
  };
  switch (option) {
    case HS_OPT_QUERY_BATCH:
      if (value < 0 || value >= (1ll << 27)) break;
      kn.query_batch = (uint32_t)value;
      return HS_OK;
    case HS_OPT_SEG_MODE:
      if (value < 0 || value > 2) break;
      kn.seg_mode = (int)value;
      return HS_OK;
    case HS_OPT_JOIN_RESIDENT:
      if (value < 0 || value > 2) break;
      kn.no_join_r = value == 1;
      kn.force_join_r = value == 2;
      return HS_OK;
    case HS_OPT_RECOGNISE_KMERS: return flag(&kn.no_recognise, true);
    case HS_OPT_BUILD_GROUPING: return flag(&kn.build_sort, false);
    case HS_OPT_WIDE_ROWS: {
      if (value < 0 || value > 3) break;
      kn.force_wide = value == 1;
      kn.no_wide_by_radius = value >= 2;
      // 3: the 4-column rows for short k-mers too -- the index's member records are built for one form
      const bool wide8 = h->wide8_ok && (int)h->p.k <= 20 && value != 3;
      if (wide8 != h->wide8) {
        drop_index(h);
        h->wide8 = wide8;
      }
      return HS_OK;
    }
    case HS_OPT_REFINE8: return flag(&kn.no_refine8, true);
    case HS_OPT_SELF_CODES: return flag(&kn.no_self_codes, true);
    case HS_OPT_SORT_HITS: return flag(&kn.sort_hits, false);
    case HS_OPT_SYNC_ITEMS: return flag(&kn.sync_items, false);
    case HS_OPT_JOIN_MIN_Q:
      if (value < 1 || value > (1ll << 30)) break;
      h->join_min_q = (uint32_t)value;
      return HS_OK;
    case HS_OPT_JOIN_MIN_M:
      if (value < 1 || value > (1ll << 30)) break;
      h->join_min_m = (uint32_t)value;
      return HS_OK;
    case HS_OPT_SORT_FROM_BIT:
      if (value < 0 || value > 60) break;
      kn.sort_from_bit = (int)value;
      return HS_OK;
    case HS_OPT_BUILD_SERIAL: return flag(&kn.build_serial, false);
    case HS_OPT_PROBE_RECORDS: return flag(&kn.no_probe_records, true);
    case HS_OPT_JOIN_CHUNK:
      if (value != 0 && (value < 2 || value > 64)) break;
      kn.join_chunk = (uint32_t)value;
      return HS_OK;
    case HS_OPT_JOIN_XCD_RUN:
      if (value < -1 || value > 4096 || (value > 0 && (value & (value - 1)))) break;  // a power of two
      kn.join_xcd_run = (int)value;
      return HS_OK;
    default:
      return fail(h, HS_ERR_INVALID, "hs_set_option: unknown option");
  }
  return fail(h, HS_ERR_INVALID, "hs_set_option: value out of range");
}

hs_status hs_wait_event(hs_handle* h, void* hip_event) {
  if (!h || !hip_event) return HS_ERR_INVALID;
  hs_status st = ensure_device(h);
  if (st) return st;
  HS_HIP(h, hipStreamWaitEvent(h->stream, reinterpret_cast<hipEvent_t>(hip_event), 0));
  return HS_OK;
}

void hs_destroy(hs_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->p.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  DevBuf* bufs[] = {&h->a, &h->aT, &h->b, &h->coords, &h->codes, &h->packed_all, &h->qints, &h->qstart,
                    &h->qcount, &h->nslices, &h->slice_off, &h->tq, &h->prov, &h->hit_key,
                    &h->hit_val, &h->hit_key2, &h->hit_val2, &h->counters, &h->temp,
                    &h->io_centers, &h->io_q, &h->io_id, &h->io_table, &h->io_dist, &h->io_cand,
                    &h->io_codes, &h->io_misc, &h->jtab, &h->c16, &h->seg_keys, &h->seg_keys_sorted,
                    &h->seg_vals, &h->sorted_ql, &h->seg_key, &h->seg_cnt, &h->seg_qoff,
                    &h->seg_items, &h->item_off, &h->seg_n, &h->c16s, &h->item_desc,
                    &h->probe_slow, &h->jtab8, &h->c8b, &h->prov2, &h->t_packed, &h->t_rec8, &h->t_rec8w, &h->t_pos, &h->dir_base,
                    &h->bucket_work, &h->proj_aq_all, &h->proj_aq_tab, &h->proj_fn, &h->proj_tab, &h->proj_stats,
                    &h->proj_flags[0], &h->proj_flags[1], &h->proj_flags[2], &h->proj_cnt, &h->proj_xq,
                    &h->proj_xmeta, &h->qhits, &h->bs_ints2[0], &h->bs_ints2[1], &h->bs_keys2[0],
                    &h->bs_keys2[1], &h->bs_iota2[0], &h->bs_iota2[1], &h->bs_keys_sorted, &h->bs_rle_unique,
                    &h->bs_rle_counts, &h->bs_small, &h->bs_sort_temp, &h->bs_slow_q, &h->all_codes,
                    &h->subset_ids, &h->qcodes_buf, &h->qembed, &h->seg_res, &h->seg_of, &h->t_rho, &h->rec_codes, &h->qpacked, &h->hit_rank, &h->hit_kv, &h->bs_fptab, &h->bs_blk,
                    &h->bs_dk, &h->bs_hist, &h->bs_rank};
  for (DevBuf* bf : bufs) bf->release();
  h->sj_host.release();
  h->t_dirjump.release();
  h->t_dirrec.release();
  h->part_work.release();
  h->t_giant.release();
  for (int l = 0; l < HS_MAX_L; ++l) {
    h->t_dirkey[l].release();
    h->t_dirstart[l].release();
    h->t_dirtuple[l].release();
    h->t_ids[l].release();
  }
  if (h->pin_cnt) (void)hipHostFree(h->pin_cnt);
  if (h->ev_ok)
    for (int i = 0; i < EV_COUNT; ++i) (void)hipEventDestroy(h->ev[i]);
  if (h->evx_ok)
    for (int i = 0; i < EVX_COUNT; ++i) (void)hipEventDestroy(h->evx[i]);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

// ------------------------------------------------------------------------------ embed / hash
hs_status hs_embed_codes(hs_handle* h, const uint8_t* codes, uint64_t n, double* out) {
  if (!h || (n && (!codes || !out))) return HS_ERR_INVALID;
  if (!n) return HS_OK;
  for (uint64_t i = 0; i < n * h->p.k; ++i)
    if (codes[i] >= h->alphabet) return fail(h, HS_ERR_INVALID, "residue code outside the alphabet");
  hs_status st = ensure_device(h);
  if (st) return st;
  const size_t out_bytes = (size_t)n * h->d * 8;
  HS_HIP(h, h->io_codes.reserve((size_t)n * h->p.k));
  HS_HIP(h, h->io_misc.reserve(out_bytes));
  HS_HIP(h, hipMemcpyAsync(h->io_codes.p, codes, (size_t)n * h->p.k, hipMemcpyHostToDevice, h->stream));
  HS_HIP(h, hs_launch_embed(h->io_codes.as<uint8_t>(), n, (int)h->p.k, h->coords.as<double>(),
                            h->io_misc.as<double>(), h->stream));
  HS_HIP(h, hipMemcpyAsync(out, h->io_misc.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  return HS_OK;
}

hs_status hs_hash_codes(hs_handle* h, const uint8_t* codes, uint64_t n, int32_t* buckets) {
  if (!h || (n && (!codes || !buckets))) return HS_ERR_INVALID;
  if (!n) return HS_OK;
  for (uint64_t i = 0; i < n * h->p.k; ++i)
    if (codes[i] >= h->alphabet) return fail(h, HS_ERR_INVALID, "residue code outside the alphabet");
  hs_status st = ensure_device(h);
  if (st) return st;
  const size_t out_bytes = (size_t)n * h->LK * 4;
  HS_HIP(h, h->io_codes.reserve((size_t)n * h->p.k));
  HS_HIP(h, h->io_misc.reserve(out_bytes));
  HS_HIP(h, hipMemcpyAsync(h->io_codes.p, codes, (size_t)n * h->p.k, hipMemcpyHostToDevice, h->stream));
  HS_HIP(h, hipEventRecord(h->ev[0], h->stream));
  HS_CHECK(hash_dispatch(h, h->io_codes.as<uint8_t>(), nullptr, n, -1, h->io_misc.as<int32_t>(), h->LK, 2,
                         h->stream));
  HS_HIP(h, hipEventRecord(h->ev[1], h->stream));
  HS_HIP(h, hipMemcpyAsync(buckets, h->io_misc.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  memset(&h->prof, 0, sizeof(h->prof));
  h->prof.ms_hash = h->prof.ms_total = ev_ms(h, 0, 1);
  return hash_account(h, n, h->LK, 2);
}

hs_status hs_hash_points(hs_handle* h, const double* points, uint64_t n, int32_t* buckets) {
  if (!h || (n && (!points || !buckets))) return HS_ERR_INVALID;
  if (!n) return HS_OK;
  hs_status st = ensure_device(h);
  if (st) return st;
  const size_t in_bytes = (size_t)n * h->d * 8, out_bytes = (size_t)n * h->LK * 4;
  HS_HIP(h, h->io_centers.reserve(in_bytes));
  HS_HIP(h, h->io_misc.reserve(out_bytes));
  HS_HIP(h, hipMemcpyAsync(h->io_centers.p, points, in_bytes, hipMemcpyHostToDevice, h->stream));
  HS_HIP(h, hipEventRecord(h->ev[0], h->stream));
  HS_CHECK(hash_dispatch(h, nullptr, h->io_centers.as<double>(), n, -1, h->io_misc.as<int32_t>(), h->LK, 2,
                         h->stream));
  HS_HIP(h, hipEventRecord(h->ev[1], h->stream));
  HS_HIP(h, hipMemcpyAsync(buckets, h->io_misc.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  memset(&h->prof, 0, sizeof(h->prof));
  h->prof.ms_hash = h->prof.ms_total = ev_ms(h, 0, 1);
  return hash_account(h, n, h->LK, 2);
}

static inline int bit_width_u32(uint32_t v) {
  int b = 0;
  while (v) {
    ++b;
    v >>= 1;
  }
  return b;
}
// bits of a sorted position inside one table (segment keys: hs_launch_seg_keys)
static inline int seg_shift_of(const hs_handle* h) {
  return std::max(1, bit_width_u32((uint32_t)std::min<uint64_t>(h->n, 0xffffffffull)));
}

// Segments -> work items of jm members x <= 2048 queries: routing (join or streaming), the item
// numbering order (many-query segments first), item offsets.  Workspace reuse: seg_keys = flags and
// their scan, seg_vals = order, seg_keys_sorted = item counts in that order (all free by now).
// max_q_res > 0: segments with at most that many probing queries form the item list's tail, whose bounds
// go to seg_n[2..3] (hs_join8r_kernel's share).
static hs_status cut_items(hs_handle* h, uint32_t nql, uint32_t jm, unsigned long long* d_jstats,
                           uint32_t max_q_res, uint32_t n_probes) {
  const size_t n1 = (size_t)nql + 1;
  // every segment with a member goes to the join (the default): no probe keeps a slice for the streaming filter
  const bool all_joined = h->join_min_q == 1 && h->join_min_m == 1;
  uint32_t* big = h->seg_keys.as<uint32_t>();
  uint32_t* big_pos = big + n1;
  uint32_t* order = h->seg_vals.as<uint32_t>();
  uint32_t* items_ord = h->seg_keys_sorted.as<uint32_t>();
  uint32_t *res = nullptr, *res_pos = nullptr;
  if (max_q_res) {
    HS_HIP(h, h->seg_res.reserve(2 * n1 * 4));
    res = h->seg_res.as<uint32_t>();
    res_pos = res + n1;
  }
  HS_HIP(h, hs_launch_seg_route(h->seg_key.as<uint64_t>(), h->seg_cnt.as<uint32_t>(),
                                h->seg_qoff.as<uint32_t>(), h->seg_n.as<uint32_t>(),
                                h->sorted_ql.as<uint32_t>(), h->qcount.as<uint32_t>(), nql,
                                h->join_min_q, h->join_min_m, jm, (int)h->p.L, seg_shift_of(h), max_q_res,
                                h->seg_items.as<uint32_t>(), d_jstats, all_joined ? nullptr : h->nslices.as<uint32_t>(),
                                h->seg_of.as<uint32_t>(), h->stream));
  if (all_joined) HS_HIP(h, hipMemsetAsync(h->nslices.p, 0, (size_t)n_probes * 4, h->stream));
  HS_HIP(h, hipMemsetAsync(big + nql, 0, 4, h->stream));
  if (res) HS_HIP(h, hipMemsetAsync(res + nql, 0, 4, h->stream));
  HS_HIP(h, hipMemsetAsync(items_ord + nql, 0, 4, h->stream));
  HS_HIP(h, hs_launch_seg_big(h->seg_cnt.as<uint32_t>(), h->seg_items.as<uint32_t>(), nql, 512u, max_q_res, big,
                              res, h->stream));
  HS_HIP(h, hs_exclusive_scan_u32(h->temp.p, h->temp.cap, big, big_pos, n1, h->stream));
  if (res) HS_HIP(h, hs_exclusive_scan_u32(h->temp.p, h->temp.cap, res, res_pos, n1, h->stream));
  HS_HIP(h, hs_launch_seg_order(big_pos, res_pos, h->seg_items.as<uint32_t>(), nql, order, items_ord,
                                h->stream));
  HS_HIP(h, hs_exclusive_scan_u32(h->temp.p, h->temp.cap, items_ord, h->item_off.as<uint32_t>(), n1,
                                  h->stream));
  HS_HIP(h, hs_launch_item_split(h->item_off.as<uint32_t>(), res_pos, nql, h->seg_n.as<uint32_t>() + 2, h->stream));
  return HS_OK;
}

// The allocations of a build whose sizes follow from n alone, made while the packing kernel (and the copy
// of the codes before it) still runs: an allocation costs ~ 0.3 ms whatever its size, and a first build
// makes two dozen of them -- with the GPU idle, they were 6.5 of its 30 ms at the C2 sizes.  build_tables
// reserves the same buffers again (no-ops then).
static hs_status reserve_build_buffers(hs_handle* h) {
  const uint64_t n = h->n;
  const int K = (int)h->p.K, L = (int)h->p.L, k = (int)h->p.k, PW = h->PW;
  for (int i = 0; i < 2; ++i) {
    HS_HIP(h, h->bs_ints2[i].reserve(std::max<size_t>(16, (size_t)n * K * 4)));
    HS_HIP(h, h->bs_keys2[i].reserve(std::max<size_t>(16, (size_t)n * 8)));
    HS_HIP(h, h->bs_iota2[i].reserve(std::max<size_t>(16, (size_t)n * 4)));
  }
  HS_HIP(h, h->bs_keys_sorted.reserve(std::max<size_t>(16, (size_t)n * 8)));
  HS_HIP(h, h->bs_rle_unique.reserve(std::max<size_t>(16, (size_t)n * 8)));
  HS_HIP(h, h->bs_rle_counts.reserve(std::max<size_t>(16, (size_t)n * 4)));
  HS_HIP(h, h->bs_small.reserve(64));
  HS_HIP(h, h->bs_sort_temp.reserve(std::max(std::max(hs_sort_pairs_u64_u32_temp(n), hs_rle_u64_temp(n)),
                                             hs_scan_u32_temp(n + 1)) + 256));
  const bool with_rec8 = h->join8_tables_ok && k <= 50;
  HS_HIP(h, h->t_packed.reserve(((size_t)L * n + HS_JM_WAVE) * PW * 16));
  if (with_rec8) HS_HIP(h, h->t_rec8.reserve(((size_t)L * n + HS_JM_WAVE) * 16));
  if (with_rec8 && k <= 25 && !h->wide8) HS_HIP(h, h->t_rho.reserve(((size_t)L * n + HS_JM_WAVE + 4) * 4));
  HS_HIP(h, h->t_pos.reserve(std::max<size_t>(16, (size_t)L * n * 4)));
  for (int l = 0; l < L; ++l) HS_HIP(h, h->t_ids[l].reserve(std::max<size_t>(16, (size_t)n * 4)));
  if (!h->knobs.build_sort && n && n < (1ull << 31)) {
    const uint32_t C = hs_group_table_slots(n), n_blk = (C + 1023) / 1024, n_tiles = hs_rs_blocks(n);
    HS_HIP(h, h->bs_fptab.reserve((size_t)C * 8));
    HS_HIP(h, h->bs_blk.reserve(2 * ((size_t)n_blk + 2) * 4));
    HS_HIP(h, h->bs_rank.reserve((size_t)n * 4));
    HS_HIP(h, h->bs_hist.reserve(2 * (size_t)((size_t)256 * n_tiles + 64) * 4));
  }
  return HS_OK;
}

// ------------------------------------------------------------------------------------- build
static hs_status build_tables(hs_handle* h, uint32_t seed, bool* collided) {
  const uint64_t n = h->n;
  const int K = (int)h->p.K, L = (int)h->p.L, k = (int)h->p.k, PW = h->PW;
  *collided = false;
  // scratch shared by all tables.  Hashing (fp64 vector ALU) and grouping (radix sort: memory) of
  // consecutive tables overlap: table l + 1 is hashed on the side stream into the other half of
  // the double-buffered ints / keys / iota while table l is sorted on the main stream.
  // (the scratch lives in the handle: Clustering() rebuilds a 10^6-k-mer index per table, and twelve
  // hipMalloc / hipFree pairs per build cost more than the build's kernels; released after the build
  // only when it is large)
  DevBuf(&ints2)[2] = h->bs_ints2, (&keys2)[2] = h->bs_keys2, (&iota2)[2] = h->bs_iota2;
  DevBuf &keys_sorted = h->bs_keys_sorted, &rle_unique = h->bs_rle_unique, &rle_counts = h->bs_rle_counts,
         &small = h->bs_small, &sort_temp = h->bs_sort_temp, &slow_q = h->bs_slow_q;
  hipEvent_t ev_hashed[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr}, ev_t[4] = {};
  // The hash runs on the handle's side stream (lowest priority: the sort's many small kernels get
  // the CUs they ask for, the hash fills the rest).  Measured at 10 M x 8 tables, repeated builds:
  // 44 ms without the overlap, 40 ms with it (the sort slows from 20 to 27 ms beside the hash); a
  // normal-priority stream gave 44 ms; a CU-masked stream (hipExtStreamCreateWithCUMask) reserving
  // a quarter of the CUs for the sort hung in the second build of a process and was dropped.
  hipStream_t hash_stream = h->stream2;
  const bool own_hash_stream = false;
  const bool serial = h->knobs.build_serial;  // measurement: no overlap
  // grouping by key: hs_group.hip (table of distinct fingerprints + a radix sort on 32-bit ranks) unless
  // HS_BUILD_SORT asks for the full-width sort of rounds 1-2 (which also remains the fallback of a table
  // the other path cannot take: a fingerprint equal to its empty marker, or nearly all keys distinct)
  const bool group_by_rank = !h->knobs.build_sort && n < (1ull << 31);
  struct Guard {
    hs_handle* h;
    hipStream_t* hs;
    bool own;
    hipEvent_t* e[3];
    int ne[3];
    DevBuf* b[17];
    ~Guard() {
      // nothing may still be running on either stream when the scratch goes away
      if (*hs) (void)hipStreamSynchronize(*hs);
      (void)hipStreamSynchronize(h->stream);
      if (*hs && own) (void)hipStreamDestroy(*hs);
      for (int g = 0; g < 3; ++g)
        for (int i = 0; i < ne[g]; ++i)
          if (e[g][i]) (void)hipEventDestroy(e[g][i]);
      size_t held = 0;
      for (DevBuf* x : b) held += x->cap;
      // The build scratch stays for the next build while it is small beside the device's memory (1/16 of
      // it: 2.2 GB at the C2 sizes stay -- freeing them took 2.5 ms of a 23 ms build --, 22 GB at the C3
      // shape go).
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)16 << 30;
      if (held > total_b / 16)
        for (DevBuf* x : b) x->release();
    }
  } guard = {h, &hash_stream, own_hash_stream, {ev_hashed, ev_free, ev_t}, {2, 2, 4},
             {&ints2[0], &ints2[1], &keys2[0], &keys2[1], &iota2[0], &iota2[1], &keys_sorted, &rle_unique,
              &rle_counts, &small, &sort_temp, &slow_q, &h->bs_fptab, &h->bs_blk, &h->bs_dk, &h->bs_hist, &h->bs_rank}};
  for (int i = 0; i < 2; ++i) {
    HS_HIP(h, hipEventCreateWithFlags(&ev_hashed[i], hipEventDisableTiming));
    HS_HIP(h, hipEventCreateWithFlags(&ev_free[i], hipEventDisableTiming));
    HS_HIP(h, ints2[i].reserve(std::max<size_t>(16, (size_t)n * K * 4)));
    HS_HIP(h, keys2[i].reserve(std::max<size_t>(16, (size_t)n * 8)));
    HS_HIP(h, iota2[i].reserve(std::max<size_t>(16, (size_t)n * 4)));
  }
  for (int i = 0; i < 4; ++i) HS_HIP(h, hipEventCreate(&ev_t[i]));  // hash start/end per buffer
  HS_HIP(h, keys_sorted.reserve(std::max<size_t>(16, (size_t)n * 8)));
  HS_HIP(h, rle_unique.reserve(std::max<size_t>(16, (size_t)n * 8)));
  HS_HIP(h, rle_counts.reserve(std::max<size_t>(16, (size_t)n * 4)));
  HS_HIP(h, small.reserve(64));  // [0]=runs [1]=collision flag [2]=max count
  const size_t temp_bytes = std::max(std::max(hs_sort_pairs_u64_u32_temp(n), hs_rle_u64_temp(n)),
                                     hs_scan_u32_temp(n + 1)) + 256;
  HS_HIP(h, sort_temp.reserve(temp_bytes));
  uint32_t* d_small = small.as<uint32_t>();
  double ms_hash = 0, ms_sort = 0, ms_gather = 0;
  const bool with_rec8 = h->join8_tables_ok && k <= 50;
  // (+ 128 entries: hs_join8r_kernel reads a bucket's ragged last member tile without clamping)
  HS_HIP(h, h->t_packed.reserve(((size_t)L * n + HS_JM_WAVE) * PW * 16));
  if (with_rec8) HS_HIP(h, h->t_rec8.reserve(((size_t)L * n + HS_JM_WAVE) * 16));
  const bool with_rho = with_rec8 && k <= 25 && !h->wide8;
  if (with_rho) HS_HIP(h, h->t_rho.reserve(((size_t)L * n + HS_JM_WAVE + 4) * 4));
  HS_HIP(h, h->t_pos.reserve(std::max<size_t>(16, (size_t)L * n * 4)));
  // hash + fingerprints of table t into buffer t & 1, on the side stream
  auto hash_table = [&](int t) -> hs_status {
    const int u = t & 1;
    if (t >= 2) HS_HIP(h, hipStreamWaitEvent(hash_stream, ev_free[u], 0));  // table t - 2 is done with it
    HS_HIP(h, hipEventRecord(ev_t[2 * u], hash_stream));
    HS_CHECK(hash_dispatch(h, h->codes.as<uint8_t>(), nullptr, n, t, ints2[u].as<int32_t>(), K, u, hash_stream));
    if (!group_by_rank)  // (hs_group.hip fingerprints the bucket ints itself, in its one pass over them)
      HS_HIP(h, hs_launch_keys(ints2[u].as<int32_t>(), n, K, K, seed, keys2[u].as<uint64_t>(),
                               iota2[u].as<uint32_t>(), hash_stream));
    HS_HIP(h, hipEventRecord(ev_t[2 * u + 1], hash_stream));
    HS_HIP(h, hipEventRecord(ev_hashed[u], hash_stream));
    return HS_OK;
  };
  {  // the side stream starts after everything queued so far (codes, packing, planes)
    HS_HIP(h, hipEventRecord(h->evx[EV_FORK], h->stream));
    HS_HIP(h, hipStreamWaitEvent(hash_stream, h->evx[EV_FORK], 0));
    HS_CHECK(hash_table(0));
  }
  for (int l = 0; l < L; ++l) {
    uint4* const tab_packed = h->t_packed.as<uint4>() + (size_t)l * n * PW;
    DevBuf& ints = ints2[l & 1];
    DevBuf& keys = keys2[l & 1];
    DevBuf& iota = iota2[l & 1];
    HS_HIP(h, h->t_ids[l].reserve(std::max<size_t>(16, (size_t)n * 4)));
    HS_HIP(h, hipMemsetAsync(d_small, 0, 64, h->stream));
    if (l + 1 < L && !serial) HS_CHECK(hash_table(l + 1));  // runs beside this table's sort
    HS_HIP(h, hipStreamWaitEvent(h->stream, ev_hashed[l & 1], 0));
    HS_HIP(h, hipEventRecord(h->ev[1], h->stream));
    uint32_t nb = 0, flag = 0, max_count = 0;
    // The radix sort looks at the fingerprints' top 48 bits only (6 passes instead of 8): buckets are
    // far fewer than 2^24, so two DISTINCT fingerprints rarely agree there (~ nb^2 / 2^49 per table);
    // hs_check_runs_kernel sees it if they do (flag 4) and the table is sorted again on all 64 bits.
    // Only where rocPRIM's onesweep passes do the sorting (hs_sort_partial_bits_ok: n above the library's
    // merge_sort_limit, 2^20): up to that size its merge sort takes over, whose cost does not depend on
    // the bits and whose comparator for a range ending at bit 64 is built from 1 << 64 (hs_prims.hip) --
    // the memory fault of this sort's first draft.  (HS_SORT_FROM_BIT: 0 = every bit at once; the tests
    // pass 56 to see the second sort happen.)
    const int from_bit0 = hs_sort_partial_bits_ok((size_t)n) ? h->knobs.sort_from_bit : 0;
    bool grouped = false;   // the table + rank-sort path produced this table's ids and directory
    if (group_by_rank && n) {
      const uint32_t C = hs_group_table_slots(n), n_blk = (C + 1023) / 1024, n_tiles = hs_rs_blocks(n);
      HS_HIP(h, h->bs_fptab.reserve((size_t)C * 8));    // the slots: 64-bit fingerprints
      HS_HIP(h, h->bs_blk.reserve(2 * ((size_t)n_blk + 2) * 4));
      HS_HIP(h, h->bs_rank.reserve((size_t)n * 4));
      HS_HIP(h, h->bs_hist.reserve(2 * (size_t)((size_t)256 * n_tiles + 64) * 4));
      uint32_t* blk_cnt = h->bs_blk.as<uint32_t>();
      uint32_t* blk_off = blk_cnt + n_blk + 2;
      uint32_t* rank = h->bs_rank.as<uint32_t>();
      HS_HIP(h, hs_launch_group_insert(ints.as<int32_t>(), n, K, seed, h->bs_fptab.as<uint64_t>(), C, rank,
                                       d_small + 1, h->stream));
      HS_HIP(h, hs_launch_fp_count(h->bs_fptab.as<uint64_t>(), C, blk_cnt, h->stream));
      HS_HIP(h, hipMemsetAsync(blk_cnt + n_blk, 0, 4, h->stream));
      HS_HIP(h, hs_exclusive_scan_u32(sort_temp.p, sort_temp.cap, blk_cnt, blk_off, (size_t)n_blk + 1, h->stream));
      uint32_t host2[2] = {0, 0};
      HS_HIP(h, hipMemcpyAsync(&host2[0], blk_off + n_blk, 4, hipMemcpyDeviceToHost, h->stream));
      HS_HIP(h, hipMemcpyAsync(&host2[1], d_small + 1, 4, hipMemcpyDeviceToHost, h->stream));
      HS_HIP(h, hipStreamSynchronize(h->stream));   // the table's one round trip: its number of buckets
#ifdef HS_TEST_HOOKS
      if (h->knobs.test_group_fallback && (l & 1)) host2[1] |= 16u;  // (every other table, so that both forms mix)
#endif
      if (h->knobs.build_debug && (host2[1] & 48u))
        fprintf(stderr, "table %d: grouping by rank not possible (flag 0x%x), sorting\n", l, host2[1]);
      if (!(host2[1] & 48u)) {
        nb = host2[0];
        grouped = true;
        HS_HIP(h, h->t_dirkey[l].reserve(std::max<size_t>(16, (size_t)nb * 8)));
        HS_HIP(h, h->t_dirstart[l].reserve(((size_t)nb + 1) * 4));
        HS_HIP(h, h->t_dirtuple[l].reserve(std::max<size_t>(16, (size_t)nb * K * 4)));
        HS_HIP(h, h->bs_dk.reserve((size_t)nb * 16 + 64));   // distinct keys (8) + their slots (4) + slots sorted (4)
        uint64_t* dk = h->bs_dk.as<uint64_t>();
        uint32_t* ds = reinterpret_cast<uint32_t*>(dk + nb);
        uint32_t* ds_sorted = ds + nb;
        HS_HIP(h, hs_launch_fp_compact(h->bs_fptab.as<uint64_t>(), C, blk_off, dk, ds, h->stream));
        HS_HIP(h, hs_sort_pairs_u64_u32(sort_temp.p, sort_temp.cap, dk, h->t_dirkey[l].as<uint64_t>(), ds, ds_sorted,
                                        nb, 0, 64, h->stream));
        // (the table's slots are free now: they take the rank of every slot's key)
        uint32_t* rank_of_slot = h->bs_fptab.as<uint32_t>();
        HS_HIP(h, hs_launch_rank_slots(ds_sorted, nb, rank_of_slot, h->stream));
        HS_HIP(h, hs_launch_rank_kmers(rank, n, rank_of_slot, h->stream));
        // stable LSD radix sort of (rank, id) on the bits the ranks have; the last pass writes the table's ids
        const int bits = std::max(1, bit_width_u32(nb ? nb - 1 : 0));
        const int n_pass = (bits + 7) / 8;
        uint32_t* kbuf[2] = {keys_sorted.as<uint32_t>(), keys_sorted.as<uint32_t>() + n};
        uint32_t* ibuf[2] = {h->t_ids[l].as<uint32_t>(), rle_counts.as<uint32_t>()};   // pass p writes ibuf[(n_pass - 1 - p) & 1]
        uint32_t* hist = h->bs_hist.as<uint32_t>();
        uint32_t* hist_scanned = hist + ((size_t)256 * n_tiles + 64);
        const uint32_t* kin = rank;
        const uint32_t* iin = nullptr;
        for (int p = 0; p < n_pass; ++p) {
          uint32_t* kout = kbuf[p & 1];
          uint32_t* iout = ibuf[(n_pass - 1 - p) & 1];
          HS_HIP(h, hs_launch_rs_hist(kin, (uint32_t)n, 8 * p, hist, h->stream));
          HS_HIP(h, hs_exclusive_scan_u32(sort_temp.p, sort_temp.cap, hist, hist_scanned, (size_t)256 * n_tiles, h->stream));
          HS_HIP(h, hs_launch_rs_scatter(kin, iin, (uint32_t)n, 8 * p, hist_scanned, kout, iout, h->stream));
          kin = kout;
          iin = iout;
        }
        HS_HIP(h, hs_launch_dir_start(kin, (uint32_t)n, nb, h->t_dirstart[l].as<uint32_t>(), d_small + 2, h->stream));
        HS_HIP(h, hs_launch_dir_tuples(h->t_dirstart[l].as<uint32_t>(), h->t_ids[l].as<uint32_t>(), ints.as<int32_t>(),
                                       nb, K, h->t_dirtuple[l].as<int32_t>(), h->stream));
        // the exact-membership proof of every k-mer against its bucket's tuple (flag 1, read with max_count below)
        HS_HIP(h, hs_launch_group_check(ints.as<int32_t>(), n, K, rank, h->t_dirtuple[l].as<int32_t>(), d_small + 1,
                                        h->stream));
      } else {
        HS_HIP(h, hipMemsetAsync(d_small, 0, 64, h->stream));
        // the sorting path needs the fingerprints and the ids 0 .. n - 1 as its values
        HS_HIP(h, hs_launch_keys(ints.as<int32_t>(), n, K, K, seed, keys.as<uint64_t>(), iota.as<uint32_t>(), h->stream));
      }
    }
    for (int from_bit = from_bit0; n && !grouped; from_bit = 0) {
      HS_HIP(h, hs_sort_pairs_u64_u32(sort_temp.p, sort_temp.cap, keys.as<uint64_t>(),
                                      keys_sorted.as<uint64_t>(), iota.as<uint32_t>(),
                                      h->t_ids[l].as<uint32_t>(), n, from_bit, 64, h->stream));
      const uint32_t slow_cap = 1u << 16;
      HS_HIP(h, slow_q.reserve(((size_t)slow_cap + 1) * 4));
      HS_HIP(h, hs_launch_check_runs(keys_sorted.as<uint64_t>(), h->t_ids[l].as<uint32_t>(),
                                     ints.as<int32_t>(), n, K, d_small + 1, slow_q.as<uint32_t>(),
                                     slow_cap, false, from_bit, h->stream));
      HS_HIP(h, hs_rle_u64(sort_temp.p, sort_temp.cap, keys_sorted.as<uint64_t>(),
                           rle_unique.as<uint64_t>(), rle_counts.as<uint32_t>(), d_small, n,
                           h->stream));
      uint32_t host_small[3];
      HS_HIP(h, hipMemcpyAsync(host_small, d_small, 8, hipMemcpyDeviceToHost, h->stream));
      HS_HIP(h, hipStreamSynchronize(h->stream));
      nb = host_small[0];
      flag = host_small[1];
      if ((flag & 4u) && from_bit) {  // interleaved fingerprints: once more, on every bit
        if (h->knobs.build_debug) fprintf(stderr, "table %d: %u buckets, second sort on all bits\n", l, nb);
        HS_HIP(h, hipMemsetAsync(d_small, 0, 64, h->stream));
        continue;
      }
      if (flag & 2u) {  // very many aliased neighbours (tiny W): compare every pair as strings
        HS_HIP(h, hipMemsetAsync(d_small + 1, 0, 4, h->stream));
        HS_HIP(h, hs_launch_check_runs(keys_sorted.as<uint64_t>(), h->t_ids[l].as<uint32_t>(),
                                       ints.as<int32_t>(), n, K, d_small + 1, slow_q.as<uint32_t>(),
                                       slow_cap, true, 0, h->stream));
        HS_HIP(h, hipMemcpyAsync(&flag, d_small + 1, 4, hipMemcpyDeviceToHost, h->stream));
        HS_HIP(h, hipStreamSynchronize(h->stream));
      }
      break;
    }
    if (flag & 1u) {
      *collided = true;
      return HS_OK;
    }
    if (!grouped) {
    HS_HIP(h, h->t_dirkey[l].reserve(std::max<size_t>(16, (size_t)nb * 8)));
    HS_HIP(h, h->t_dirstart[l].reserve(((size_t)nb + 1) * 4));
    HS_HIP(h, h->t_dirtuple[l].reserve(std::max<size_t>(16, (size_t)nb * K * 4)));
    if (nb) {
      HS_HIP(h, hipMemcpyAsync(h->t_dirkey[l].p, rle_unique.p, (size_t)nb * 8,
                               hipMemcpyDeviceToDevice, h->stream));
      HS_HIP(h, hs_exclusive_scan_u32(sort_temp.p, sort_temp.cap, rle_counts.as<uint32_t>(),
                                      h->t_dirstart[l].as<uint32_t>(), nb, h->stream));
      HS_HIP(h, hs_launch_max_u32(rle_counts.as<uint32_t>(), nb, d_small + 2, h->stream));
    }
    HS_HIP(h, hs_launch_set_u32(h->t_dirstart[l].as<uint32_t>() + nb, (uint32_t)n, h->stream));
    HS_HIP(h, hs_launch_dir_tuples(h->t_dirstart[l].as<uint32_t>(), h->t_ids[l].as<uint32_t>(),
                                   ints.as<int32_t>(), nb, K, h->t_dirtuple[l].as<int32_t>(),
                                   h->stream));
    }
    HS_HIP(h, hipEventRecord(ev_free[l & 1], h->stream));  // ints / keys / iota of this table are free
    HS_HIP(h, hipEventRecord(h->ev[2], h->stream));
    if (with_rec8)
      HS_HIP(h, hs_launch_gather_rec8(h->packed_all.as<uint4>(), h->t_ids[l].as<uint32_t>(), (uint32_t)n,
                                      k, h->wide8, h->jtab8.p, h->jtab8.as<char>() + 1536,
                                      h->jtab8.as<float>() + 128, tab_packed,
                                      h->t_rec8.as<uint4>() + (size_t)l * n,
                                      with_rho ? h->t_rho.as<uint32_t>() + (size_t)l * n : nullptr, h->stream));
    else
      HS_HIP(h, hs_launch_gather_packed(h->packed_all.as<uint4>(), h->t_ids[l].as<uint32_t>(), n, PW,
                                        tab_packed, h->stream));
    HS_HIP(h, hs_launch_invert_perm(h->t_ids[l].as<uint32_t>(), (uint32_t)n,
                                    h->t_pos.as<uint32_t>() + (size_t)l * n, h->stream));
    HS_HIP(h, hipEventRecord(h->ev[3], h->stream));
    uint32_t tail2[2] = {0, 0};  // {collision flag, largest bucket}
    HS_HIP(h, hipMemcpyAsync(tail2, d_small + 1, 8, hipMemcpyDeviceToHost, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    max_count = tail2[1];
    if (grouped && (tail2[0] & 1u)) {  // one fingerprint, two HashKey strings: the caller tries the next seed
      *collided = true;
      return HS_OK;
    }
    if (l + 1 < L && serial) HS_CHECK(hash_table(l + 1));
    {
      float ms = 0.f;  // the hash ran on the side stream, possibly beside the previous table's sort
      if (hipEventElapsedTime(&ms, ev_t[2 * (l & 1)], ev_t[2 * (l & 1) + 1]) == hipSuccess) ms_hash += ms;
    }
    ms_sort += ev_ms(h, 1, 2);
    ms_gather += ev_ms(h, 2, 3);
    HS_CHECK(hash_account(h, n, K, l & 1));  // table l's hash has completed (its sort waited for it)
    hs_table_dev& tb = h->tabs.t[l];
    tb.dir_key = h->t_dirkey[l].as<uint64_t>();
    tb.dir_start = h->t_dirstart[l].as<uint32_t>();
    tb.dir_tuple = h->t_dirtuple[l].as<int32_t>();
    tb.packed = tab_packed;
    tb.ids = h->t_ids[l].as<uint32_t>();
    tb.pos_of = h->t_pos.as<uint32_t>() + (size_t)l * n;
    tb.nb = nb;
    h->info.n_buckets[l] = nb;
    h->info.max_bucket[l] = max_count;
  }
  h->prof.ms_hash = ms_hash;
  h->prof.ms_sort = ms_sort;
  h->prof.ms_gather = ms_gather;
  return HS_OK;
}

// the tables as the probe kernel gets them: without the directory records when the option says so
static hs_tables_dev probe_tabs(const hs_handle* h, uint32_t q_first, const uint32_t* probe_list = nullptr,
                                uint32_t n_list = 0) {
  hs_tables_dev t = h->tabs;
  t.q_first = q_first;
  t.probe_list = probe_list;
  t.n_list = n_list;
  if (h->knobs.no_probe_records)
    for (int l = 0; l < HS_MAX_L; ++l) t.t[l].dir_rec = nullptr;
  t.part = h->self_first == HS_NO_SELF ? h->bucket_part : 0u;  // (searches only, not the self-joins)
  t.n_parts = h->self_first == HS_NO_SELF ? h->bucket_parts : 1u;
  return t;
}

// Common end of hs_index_build* and hs_index_load: info, global bucket numbering, byte count.
static hs_status finish_index(hs_handle* h) {
  h->info.n = h->n;
  h->info.key_seed = h->key_seed;
  {  // global bucket numbering over the tables (grouping of probes by bucket at query time)
    uint32_t base[HS_MAX_L + 1];
    uint64_t acc = 0;
    for (uint32_t l = 0; l < h->p.L; ++l) {
      base[l] = (uint32_t)acc;
      acc += h->info.n_buckets[l];
    }
    if (acc >= 0xfffffff0ull) return fail(h, HS_ERR_INVALID, "too many buckets");
    base[h->p.L] = (uint32_t)acc;
    h->nb_total = (uint32_t)acc;
    HS_HIP(h, h->dir_base.reserve((HS_MAX_L + 1) * 4));
    HS_HIP(h, hipMemcpyAsync(h->dir_base.p, base, ((size_t)h->p.L + 1) * 4, hipMemcpyHostToDevice, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));  // (base[] is a local)
  }
  // jump tables of the directories: the top J bits of a fingerprint (2^J >= number of buckets); one
  // allocation for all tables (an allocation costs ~ 0.5 ms, whatever its size)
  size_t jump_words = 0, jump_at[HS_MAX_L];
  for (uint32_t l = 0; l < h->p.L; ++l) {
    const uint32_t J = (uint32_t)std::max(1, bit_width_u32((uint32_t)h->info.n_buckets[l]));
    jump_at[l] = jump_words;
    jump_words += ((size_t)1 << J) + 2;
  }
  HS_HIP(h, h->t_dirjump.reserve(jump_words * 4));
  for (uint32_t l = 0; l < h->p.L; ++l) {
    const uint32_t nb = (uint32_t)h->info.n_buckets[l];
    const uint32_t J = (uint32_t)std::max(1, bit_width_u32(nb));
    const uint32_t n_slots = 1u << J;
    uint32_t* const jump = h->t_dirjump.as<uint32_t>() + jump_at[l];
    HS_HIP(h, hs_launch_dir_jump(h->t_dirkey[l].as<uint64_t>(), nb, 64 - J, n_slots, jump, h->stream));
    h->tabs.t[l].dir_jump = jump;
    h->tabs.t[l].jump_shift = 64 - J;
  }
  // the giant buckets of every table (bucket partition: shared among the parts by query, hs_probe_part)
  {
    constexpr uint32_t GCAP = 1024;  // per table: buckets of > n / 1024 members number < 1024
    HS_HIP(h, h->t_giant.reserve((size_t)HS_MAX_L * GCAP * 8));
    HS_HIP(h, h->counters.reserve(256));
    uint32_t* const d_ng = h->counters.as<uint32_t>() + 32;
    HS_HIP(h, hipMemsetAsync(d_ng, 0, HS_MAX_L * 4, h->stream));
    const uint32_t thr = hs_giant_threshold(h->n);
    for (uint32_t l = 0; l < h->p.L; ++l)
      HS_HIP(h, hs_launch_giant_buckets(h->t_dirtuple[l].as<int32_t>(), (int)h->p.K, h->t_dirstart[l].as<uint32_t>(),
                                        (uint32_t)h->info.n_buckets[l], thr, h->t_giant.as<uint64_t>() + (size_t)l * GCAP,
                                        GCAP, d_ng + l, h->stream));
    uint32_t ng[HS_MAX_L];
    std::vector<uint64_t> keys((size_t)HS_MAX_L * GCAP);
    HS_HIP(h, hipMemcpyAsync(ng, d_ng, HS_MAX_L * 4, hipMemcpyDeviceToHost, h->stream));
    HS_HIP(h, hipMemcpyAsync(keys.data(), h->t_giant.p, keys.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    for (uint32_t l = 0; l < h->p.L; ++l) {
      if (ng[l] > GCAP) return fail(h, HS_ERR_STATE, "more giant buckets in a table than its list holds");
      std::sort(keys.begin() + (size_t)l * GCAP, keys.begin() + (size_t)l * GCAP + ng[l]);
      h->tabs.t[l].giant_key = h->t_giant.as<uint64_t>() + (size_t)l * GCAP;
      h->tabs.t[l].n_giant = ng[l];
    }
    HS_HIP(h, hipMemcpyAsync(h->t_giant.p, keys.data(), keys.size() * 8, hipMemcpyHostToDevice, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));  // (keys is a local)
  }
  // directory records (hs_dir_records_kernel): one 64-byte line per bucket for the probe, where the tuples fit
  for (uint32_t l = 0; l < h->p.L; ++l) h->tabs.t[l].dir_rec = nullptr;
  if (h->p.K <= HS_REC_MAX_K && h->nb_total) {
    HS_HIP(h, h->t_dirrec.reserve((size_t)h->nb_total * 64));
    HS_HIP(h, h->counters.reserve(256));
    uint32_t* const d_wide = h->counters.as<uint32_t>() + 32;  // one flag per table (L <= 32)
    HS_HIP(h, hipMemsetAsync(d_wide, 0, HS_MAX_L * 4, h->stream));
    uint64_t at = 0;
    for (uint32_t l = 0; l < h->p.L; ++l) {
      const uint32_t nb = (uint32_t)h->info.n_buckets[l];
      uint4* const rec = h->t_dirrec.as<uint4>() + 4 * at;
      HS_HIP(h, hs_launch_dir_records(h->t_dirkey[l].as<uint64_t>(), h->t_dirstart[l].as<uint32_t>(),
                                      h->t_dirtuple[l].as<int32_t>(), nb, (int)h->p.K, rec, d_wide + l, h->stream));
      h->tabs.t[l].dir_rec = rec;
      at += nb;
    }
    uint32_t wide[HS_MAX_L];
    HS_HIP(h, hipMemcpyAsync(wide, d_wide, HS_MAX_L * 4, hipMemcpyDeviceToHost, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    for (uint32_t l = 0; l < h->p.L; ++l)
      if (wide[l]) h->tabs.t[l].dir_rec = nullptr;  // a bucket int outside 16 bits: this table keeps the arrays
  }
  HS_HIP(h, hipStreamSynchronize(h->stream));
  uint64_t bytes = h->codes.cap + h->packed_all.cap + h->t_packed.cap + h->t_rec8.cap + h->t_rec8w.cap + h->t_rho.cap +
                   h->t_pos.cap + h->t_dirrec.cap;
  for (uint32_t l = 0; l < h->p.L; ++l)
    bytes += h->t_dirkey[l].cap + h->t_dirstart[l].cap + h->t_dirtuple[l].cap + h->t_ids[l].cap;
  bytes += h->t_dirjump.cap;
  h->info.device_bytes = bytes;
  h->built = true;
  return HS_OK;
}

// Index build over the n x k residue codes already in h->codes (device): validation + packing,
// the L tables, the global bucket numbering.
static hs_status index_build_resident(hs_handle* h, uint64_t n) {
  const int k = (int)h->p.k;
  hs_status st = HS_OK;
  HS_HIP(h, h->packed_all.reserve(std::max<size_t>(16, (size_t)n * h->PW * 16)));
  if (n) {
    HS_HIP(h, hipMemsetAsync(h->counters.p, 0, 256, h->stream));
    HS_HIP(h, hs_launch_pack(h->codes.as<uint8_t>(), n, k, h->alphabet, h->packed_all.as<uint4>(),
                             h->counters.as<uint32_t>(), h->stream));
    HS_CHECK(reserve_build_buffers(h));  // (beside the copy of the codes and the packing kernel)
    uint32_t bad = 0;
    HS_HIP(h, hipMemcpyAsync(&bad, h->counters.p, 4, hipMemcpyDeviceToHost, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    if (bad) return fail(h, HS_ERR_INVALID, "residue code outside the alphabet in the DB");
  }
  bool collided = true;
  uint32_t seed = 0;
  for (; seed < 4 && collided; ++seed) {
    st = build_tables(h, seed, &collided);
    if (st) return st;
    if (!collided) break;
  }
  if (collided) return fail(h, HS_ERR_KEY_COLLISION, "key fingerprints collided for 4 seeds");
  h->key_seed = seed;
  HS_HIP(h, hipEventRecord(h->ev[9], h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  h->prof.ms_total = ev_ms(h, 8, 9);
  return finish_index(h);
}

hs_status hs_index_build(hs_handle* h, const uint8_t* codes, uint64_t n) {
  if (!h || (n && !codes)) return HS_ERR_INVALID;
  if (n >= (1ull << 31)) return fail(h, HS_ERR_INVALID, "n must be < 2^31 (ids are 32-bit, as in the reference)");
  hs_status st = ensure_device(h);
  if (st) return st;
  drop_index(h);
  h->n = n;
  memset(&h->prof, 0, sizeof(h->prof));
  memset(&h->info, 0, sizeof(h->info));
  const int k = (int)h->p.k;
  HS_HIP(h, h->codes.reserve(std::max<size_t>(16, (size_t)n * k)));
  HS_HIP(h, h->counters.reserve(256));
  HS_HIP(h, hipEventRecord(h->ev[8], h->stream));
  if (n) HS_HIP(h, hipMemcpyAsync(h->codes.p, codes, (size_t)n * k, hipMemcpyHostToDevice, h->stream));
  return index_build_resident(h, n);
}

// ---- index build with the hashing spread over ranks ("Index build" row) --------------
// The index is replicated, so every rank holds all n k-mers; what is spread is the evaluation of the
// L x K hash functions (the matrix-core part of the build): rank r does it for its contiguous block of
// the k-mers (hs_shard_bounds' rule) and the ranks exchange 8-byte fingerprints instead.  Per table:
//   hs_index_shard_hash_dev    bucket ints + fingerprints of the rank's block
//   <all-gather of the fingerprints, blocks in rank order = id order>
//   hs_index_shard_group_dev   every rank groups all n fingerprints (sort, directory)
//   hs_index_shard_tuples_dev  the bucket ints of the buckets whose FIRST member the rank hashed
//   <sum over ranks: every bucket's tuple>
//   hs_index_shard_finish_dev  exact-membership proof of the rank's own k-mers against the tuples, the
//                              bucket-ordered copies; *collided = one fingerprint, two HashKey strings
//   <max over ranks of collided: if set, every rank starts over with seed + 1>
// then hs_index_shard_end.  The index is the one hs_index_build builds, bit for bit.
// The sharded build's scratch (sorted fingerprints, run lengths, ids, sort space, the block's bucket ints:
// ~ 40 bytes per k-mer) stays with the handle for the next build while it is small beside the device's
// memory and is given back otherwise: build_tables' rule (4-5 GB beside a 157 GB index at 10^8 k-mers).
static void release_large_shard_scratch(hs_handle* h) {
  DevBuf* b[] = {&h->bs_keys_sorted, &h->bs_rle_unique, &h->bs_rle_counts, &h->bs_iota2[0], &h->bs_sort_temp,
                 &h->bs_ints2[0]};
  size_t held = 0, free_b = 0, total_b = 0;
  for (DevBuf* x : b) held += x->cap;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)16 << 30;
  if (held > total_b / 16)
    for (DevBuf* x : b) x->release();
}

hs_status hs_index_shard_begin(hs_handle* h, const uint8_t* codes, uint64_t n, uint32_t rank, uint32_t world,
                               uint64_t* block_lo, uint64_t* block_count) {
  if (!h || (n && !codes) || !world || rank >= world) return HS_ERR_INVALID;
  if (n >= (1ull << 31)) return fail(h, HS_ERR_INVALID, "n must be < 2^31 (ids are 32-bit, as in the reference)");
  hs_status st = ensure_device(h);
  if (st) return st;
  if (h->shard_open) {  // a sharded build that never reached hs_index_shard_end (a collision on every seed)
    HS_HIP(h, hipStreamSynchronize(h->stream));
    release_large_shard_scratch(h);
    h->shard_open = false;
  }
  drop_index(h);
  h->n = n;
  memset(&h->prof, 0, sizeof(h->prof));
  memset(&h->info, 0, sizeof(h->info));
  const int k = (int)h->p.k, K = (int)h->p.K, L = (int)h->p.L, PW = h->PW;
  HS_HIP(h, h->codes.reserve(std::max<size_t>(16, (size_t)n * k)));
  HS_HIP(h, h->counters.reserve(256));
  HS_HIP(h, hipEventRecord(h->ev[8], h->stream));
  if (n) HS_HIP(h, hipMemcpyAsync(h->codes.p, codes, (size_t)n * k, hipMemcpyHostToDevice, h->stream));
  HS_HIP(h, h->packed_all.reserve(std::max<size_t>(16, (size_t)n * PW * 16)));
  if (n) {
    HS_HIP(h, hipMemsetAsync(h->counters.p, 0, 256, h->stream));
    HS_HIP(h, hs_launch_pack(h->codes.as<uint8_t>(), n, k, h->alphabet, h->packed_all.as<uint4>(),
                             h->counters.as<uint32_t>(), h->stream));
    uint32_t bad = 0;
    HS_HIP(h, hipMemcpyAsync(&bad, h->counters.p, 4, hipMemcpyDeviceToHost, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    if (bad) return fail(h, HS_ERR_INVALID, "residue code outside the alphabet in the DB");
  }
  const uint64_t base = n / world, rem = n % world;   // = hs_shard_bounds (hsearch_dist.h)
  const uint64_t lo = (uint64_t)rank * base + std::min<uint64_t>(rank, rem), cnt = base + (rank < rem ? 1 : 0);
  h->shard_lo = (uint32_t)lo;
  h->shard_cnt = (uint32_t)cnt;
  h->shard_open = true;
  h->shard_table = -1;
  if (block_lo) *block_lo = lo;
  if (block_count) *block_count = cnt;
  const bool with_rec8 = h->join8_tables_ok && k <= 50;
  HS_HIP(h, h->t_packed.reserve(((size_t)L * n + HS_JM_WAVE) * PW * 16));
  if (with_rec8) HS_HIP(h, h->t_rec8.reserve(((size_t)L * n + HS_JM_WAVE) * 16));
  if (with_rec8 && k <= 25 && !h->wide8) HS_HIP(h, h->t_rho.reserve(((size_t)L * n + HS_JM_WAVE + 4) * 4));
  HS_HIP(h, h->t_pos.reserve(std::max<size_t>(16, (size_t)L * n * 4)));
  HS_HIP(h, h->bs_ints2[0].reserve(std::max<size_t>(16, (size_t)cnt * K * 4)));
  HS_HIP(h, h->bs_iota2[0].reserve(std::max<size_t>(16, (size_t)n * 4)));
  HS_HIP(h, h->bs_keys_sorted.reserve(std::max<size_t>(16, (size_t)n * 8)));
  HS_HIP(h, h->bs_rle_unique.reserve(std::max<size_t>(16, (size_t)n * 8)));
  HS_HIP(h, h->bs_rle_counts.reserve(std::max<size_t>(16, (size_t)n * 4)));
  HS_HIP(h, h->bs_small.reserve(64));
  HS_HIP(h, h->bs_sort_temp.reserve(std::max(std::max(hs_sort_pairs_u64_u32_temp(n), hs_rle_u64_temp(n)),
                                             hs_scan_u32_temp(n + 1)) + 256));
  return HS_OK;
}

hs_status hs_index_shard_hash_dev(hs_handle* h, uint32_t l, uint32_t seed, uint64_t* d_fp_block) {
  if (!h || !h->shard_open || l >= h->p.L || (h->shard_cnt && !d_fp_block)) return HS_ERR_INVALID;
  hs_status st = ensure_device(h);
  if (st) return st;
  const int K = (int)h->p.K;
  h->shard_table = (int)l;
  h->shard_seed = seed;
  if (!h->shard_cnt) return HS_OK;
  HS_HIP(h, hipEventRecord(h->ev[0], h->stream));
  HS_CHECK(hash_dispatch(h, h->codes.as<uint8_t>() + (uint64_t)h->shard_lo * h->p.k, nullptr, h->shard_cnt, (int)l,
                         h->bs_ints2[0].as<int32_t>(), K, 0, h->stream));
  HS_HIP(h, hs_launch_keys(h->bs_ints2[0].as<int32_t>(), h->shard_cnt, K, K, seed, d_fp_block, nullptr, h->stream));
  HS_HIP(h, hipEventRecord(h->ev[1], h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  h->prof.ms_hash += ev_ms(h, 0, 1);
  return hash_account(h, h->shard_cnt, K, 0);
}

hs_status hs_index_shard_group_dev(hs_handle* h, uint32_t l, const uint64_t* d_fp_all, uint32_t* n_buckets) {
  if (!h || !h->shard_open || (int)l != h->shard_table || !n_buckets || (h->n && !d_fp_all)) return HS_ERR_INVALID;
  hs_status st = ensure_device(h);
  if (st) return st;
  const uint64_t n = h->n;
  const int K = (int)h->p.K;
  *n_buckets = 0;
  uint32_t* d_small = h->bs_small.as<uint32_t>();
  HS_HIP(h, h->t_ids[l].reserve(std::max<size_t>(16, (size_t)n * 4)));
  HS_HIP(h, hipMemsetAsync(d_small, 0, 64, h->stream));
  uint32_t nb = 0;
  if (n) {
    HS_HIP(h, hipEventRecord(h->ev[1], h->stream));
    HS_HIP(h, hs_launch_iota_u32(h->bs_iota2[0].as<uint32_t>(), (uint32_t)n, h->stream));
    HS_HIP(h, hs_sort_pairs_u64_u32(h->bs_sort_temp.p, h->bs_sort_temp.cap, d_fp_all, h->bs_keys_sorted.as<uint64_t>(),
                                    h->bs_iota2[0].as<uint32_t>(), h->t_ids[l].as<uint32_t>(), n, 0, 64, h->stream));
    HS_HIP(h, hs_rle_u64(h->bs_sort_temp.p, h->bs_sort_temp.cap, h->bs_keys_sorted.as<uint64_t>(),
                         h->bs_rle_unique.as<uint64_t>(), h->bs_rle_counts.as<uint32_t>(), d_small, n, h->stream));
    HS_HIP(h, hipMemcpyAsync(&nb, d_small, 4, hipMemcpyDeviceToHost, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
  }
  HS_HIP(h, h->t_dirkey[l].reserve(std::max<size_t>(16, (size_t)nb * 8)));
  HS_HIP(h, h->t_dirstart[l].reserve(((size_t)nb + 1) * 4));
  HS_HIP(h, h->t_dirtuple[l].reserve(std::max<size_t>(16, (size_t)nb * K * 4)));
  if (nb) {
    HS_HIP(h, hipMemcpyAsync(h->t_dirkey[l].p, h->bs_rle_unique.p, (size_t)nb * 8, hipMemcpyDeviceToDevice, h->stream));
    HS_HIP(h, hs_exclusive_scan_u32(h->bs_sort_temp.p, h->bs_sort_temp.cap, h->bs_rle_counts.as<uint32_t>(),
                                    h->t_dirstart[l].as<uint32_t>(), nb, h->stream));
    HS_HIP(h, hs_launch_max_u32(h->bs_rle_counts.as<uint32_t>(), nb, d_small + 2, h->stream));
  }
  HS_HIP(h, hs_launch_set_u32(h->t_dirstart[l].as<uint32_t>() + nb, (uint32_t)n, h->stream));
  HS_HIP(h, hipEventRecord(h->ev[2], h->stream));
  uint32_t max_count = 0;
  HS_HIP(h, hipMemcpyAsync(&max_count, d_small + 2, 4, hipMemcpyDeviceToHost, h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  if (n) h->prof.ms_sort += ev_ms(h, 1, 2);
  h->shard_nb = nb;
  h->info.n_buckets[l] = nb;
  h->info.max_bucket[l] = max_count;
  *n_buckets = nb;
  return HS_OK;
}

hs_status hs_index_shard_tuples_dev(hs_handle* h, uint32_t l, int32_t* d_tuples) {
  if (!h || !h->shard_open || (int)l != h->shard_table || (h->shard_nb && !d_tuples)) return HS_ERR_INVALID;
  hs_status st = ensure_device(h);
  if (st) return st;
  HS_HIP(h, hs_launch_shard_first_tuples(h->t_dirstart[l].as<uint32_t>(), h->t_ids[l].as<uint32_t>(),
                                         h->bs_ints2[0].as<int32_t>(), h->shard_lo, h->shard_cnt, h->shard_nb,
                                         (int)h->p.K, d_tuples, h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  return HS_OK;
}

hs_status hs_index_shard_finish_dev(hs_handle* h, uint32_t l, const int32_t* d_tuples_all, uint32_t* collided) {
  if (!h || !h->shard_open || (int)l != h->shard_table || !collided || (h->shard_nb && !d_tuples_all))
    return HS_ERR_INVALID;
  hs_status st = ensure_device(h);
  if (st) return st;
  const uint64_t n = h->n;
  const int K = (int)h->p.K, k = (int)h->p.k, PW = h->PW;
  const uint32_t nb = h->shard_nb;
  uint32_t* d_small = h->bs_small.as<uint32_t>();
  uint4* const tab_packed = h->t_packed.as<uint4>() + (size_t)l * n * PW;
  const bool with_rec8 = h->join8_tables_ok && k <= 50;
  *collided = 0;
  if (nb) HS_HIP(h, hipMemcpyAsync(h->t_dirtuple[l].p, d_tuples_all, (size_t)nb * K * 4, hipMemcpyDeviceToDevice, h->stream));
  HS_HIP(h, hipEventRecord(h->ev[2], h->stream));
  if (n) {
    HS_HIP(h, hs_launch_invert_perm(h->t_ids[l].as<uint32_t>(), (uint32_t)n, h->t_pos.as<uint32_t>() + (size_t)l * n,
                                    h->stream));
    HS_HIP(h, hipMemsetAsync(d_small + 1, 0, 4, h->stream));
    HS_HIP(h, hs_launch_shard_check(h->bs_ints2[0].as<int32_t>(), h->shard_lo, h->shard_cnt, K,
                                    h->t_pos.as<uint32_t>() + (size_t)l * n, h->t_dirstart[l].as<uint32_t>(), nb,
                                    h->t_dirtuple[l].as<int32_t>(), d_small + 1, h->stream));
    if (with_rec8)
      HS_HIP(h, hs_launch_gather_rec8(h->packed_all.as<uint4>(), h->t_ids[l].as<uint32_t>(), (uint32_t)n, k, h->wide8,
                                      h->jtab8.p, h->jtab8.as<char>() + 1536, h->jtab8.as<float>() + 128, tab_packed,
                                      h->t_rec8.as<uint4>() + (size_t)l * n,
                                      (k <= 25 && !h->wide8) ? h->t_rho.as<uint32_t>() + (size_t)l * n : nullptr, h->stream));
    else
      HS_HIP(h, hs_launch_gather_packed(h->packed_all.as<uint4>(), h->t_ids[l].as<uint32_t>(), n, PW, tab_packed,
                                        h->stream));
  }
  HS_HIP(h, hipEventRecord(h->ev[3], h->stream));
  uint32_t flag = 0;
  HS_HIP(h, hipMemcpyAsync(&flag, d_small + 1, 4, hipMemcpyDeviceToHost, h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  h->prof.ms_gather += ev_ms(h, 2, 3);
  *collided = flag & 1u;
  hs_table_dev& tb = h->tabs.t[l];
  tb.dir_key = h->t_dirkey[l].as<uint64_t>();
  tb.dir_start = h->t_dirstart[l].as<uint32_t>();
  tb.dir_tuple = h->t_dirtuple[l].as<int32_t>();
  tb.packed = tab_packed;
  tb.ids = h->t_ids[l].as<uint32_t>();
  tb.pos_of = h->t_pos.as<uint32_t>() + (size_t)l * n;
  tb.nb = nb;
  h->shard_table = -1;
  return HS_OK;
}

hs_status hs_index_shard_end(hs_handle* h, uint32_t key_seed) {
  if (!h || !h->shard_open) return HS_ERR_INVALID;
  hs_status st = ensure_device(h);
  if (st) return st;
  h->shard_open = false;
  HS_HIP(h, hipStreamSynchronize(h->stream));
  release_large_shard_scratch(h);
  h->key_seed = key_seed;
  HS_HIP(h, hipEventRecord(h->ev[9], h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  h->prof.ms_total = ev_ms(h, 8, 9);
  return finish_index(h);
}

hs_status hs_index_build_subset(hs_handle* h, const uint8_t* codes_all, uint64_t n_all,
                                const uint32_t* subset, uint64_t n_subset) {
  if (!h || (n_all && !codes_all) || (!subset && n_subset != n_all)) return HS_ERR_INVALID;
  if (n_all >= (1ull << 31) || n_subset > n_all)
    return fail(h, HS_ERR_INVALID, "n must be < 2^31 and the subset no larger than the array");
  hs_status st = ensure_device(h);
  if (st) return st;
  if (subset)
    for (uint64_t i = 0; i < n_subset; ++i)
      if (subset[i] >= n_all) return fail(h, HS_ERR_INVALID, "subset index outside the code array");
  const int k = (int)h->p.k;
  drop_index(h);
  h->n = n_subset;
  memset(&h->prof, 0, sizeof(h->prof));
  memset(&h->info, 0, sizeof(h->info));
  HS_HIP(h, h->counters.reserve(256));
  HS_HIP(h, hipEventRecord(h->ev[8], h->stream));
  if (h->all_codes_key != codes_all || h->all_codes_n != n_all) {
    h->all_codes_key = nullptr;
    HS_HIP(h, h->all_codes.reserve(std::max<size_t>(16, (size_t)n_all * k)));
    if (n_all) HS_HIP(h, hipMemcpyAsync(h->all_codes.p, codes_all, (size_t)n_all * k, hipMemcpyHostToDevice, h->stream));
    h->all_codes_key = codes_all;
    h->all_codes_n = n_all;
  }
  HS_HIP(h, h->codes.reserve(std::max<size_t>(16, (size_t)n_subset * k)));
  const uint32_t* d_sub = nullptr;
  if (subset && n_subset) {
    HS_HIP(h, h->subset_ids.reserve((size_t)n_subset * 4));
    HS_HIP(h, hipMemcpyAsync(h->subset_ids.p, subset, (size_t)n_subset * 4, hipMemcpyHostToDevice, h->stream));
    d_sub = h->subset_ids.as<uint32_t>();
  }
  HS_HIP(h, hs_launch_gather_rows(h->all_codes.as<uint8_t>(), d_sub, n_subset, k, h->codes.as<uint8_t>(), h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));  // `subset` is the caller's again
  return index_build_resident(h, n_subset);
}

// DB = every length-k window of every sequence of a concatenated residue buffer (kmer_search.cpp:
// 64-83 enumerates them the same way: sequence-major, ascending offset; windows do not cross
// sequence boundaries, sequences shorter than k contribute none).  The buffer crosses PCIe once
// (n_residues bytes instead of n_windows * k) and the windows are expanded on the device.
hs_status hs_index_build_windows(hs_handle* h, const uint8_t* residues, uint64_t n_residues,
                                 const uint64_t* seq_start, uint64_t n_seq, uint64_t* n_windows,
                                 uint32_t* window_pos) {
  if (!h || !n_windows || (n_seq && !seq_start) || (n_residues && !residues)) return HS_ERR_INVALID;
  *n_windows = 0;
  hs_status st = ensure_device(h);
  if (st) return st;
  const uint64_t k = h->p.k;
  if (n_residues >= (1ull << 32)) return fail(h, HS_ERR_INVALID, "n_residues must be < 2^32");
  // first window number of every sequence
  std::vector<uint32_t> starts((size_t)n_seq + 1), win_off((size_t)n_seq + 1);
  uint64_t n = 0;
  for (uint64_t s = 0; s < n_seq; ++s) {
    if (seq_start[s] > seq_start[s + 1] || seq_start[s + 1] > n_residues)
      return fail(h, HS_ERR_INVALID, "seq_start must be ascending and end at n_residues");
    const uint64_t len = seq_start[s + 1] - seq_start[s];
    starts[s] = (uint32_t)seq_start[s];
    win_off[s] = (uint32_t)n;
    n += len >= k ? len - k + 1 : 0;
    if (n >= (1ull << 31)) return fail(h, HS_ERR_INVALID, "more than 2^31 - 1 windows");
  }
  starts[n_seq] = (uint32_t)(n_seq ? seq_start[n_seq] : 0);
  win_off[n_seq] = (uint32_t)n;
  drop_index(h);
  h->n = n;
  memset(&h->prof, 0, sizeof(h->prof));
  memset(&h->info, 0, sizeof(h->info));
  HS_HIP(h, h->codes.reserve(std::max<size_t>(16, (size_t)n * k)));
  HS_HIP(h, h->counters.reserve(256));
  HS_HIP(h, hipEventRecord(h->ev[8], h->stream));
  if (n) {
    DevBuf d_res, d_starts, d_off, d_pos;
    struct Guard {
      DevBuf* b[4];
      ~Guard() { for (DevBuf* x : b) x->release(); }
    } guard = {{&d_res, &d_starts, &d_off, &d_pos}};
    HS_HIP(h, d_res.reserve((size_t)n_residues));
    HS_HIP(h, d_starts.reserve(((size_t)n_seq + 1) * 4));
    HS_HIP(h, d_off.reserve(((size_t)n_seq + 1) * 4));
    HS_HIP(h, d_pos.reserve((size_t)n * 4));
    HS_HIP(h, hipMemcpyAsync(d_res.p, residues, (size_t)n_residues, hipMemcpyHostToDevice, h->stream));
    HS_HIP(h, hipMemcpyAsync(d_starts.p, starts.data(), ((size_t)n_seq + 1) * 4, hipMemcpyHostToDevice, h->stream));
    HS_HIP(h, hipMemcpyAsync(d_off.p, win_off.data(), ((size_t)n_seq + 1) * 4, hipMemcpyHostToDevice, h->stream));
    HS_HIP(h, hs_launch_windows(d_res.as<uint8_t>(), (uint32_t)n_residues, d_starts.as<uint32_t>(),
                                d_off.as<uint32_t>(), (uint32_t)n_seq, (int)k, h->codes.as<uint8_t>(),
                                d_pos.as<uint32_t>(), h->stream));
    if (window_pos)
      HS_HIP(h, hipMemcpyAsync(window_pos, d_pos.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
  }
  *n_windows = n;
  return index_build_resident(h, n);
}

// ---- KLSH pre-grouping ------------------------------------------------------
hs_status hs_klsh_draw_planes(uint32_t feat, uint32_t bits, double sigma, double* w, double* b,
                              double* t) {
  if (!feat || !bits || bits > 64 || !w || !b || !t) return HS_ERR_INVALID;
  // KLSH's members in declaration order (lsh.hpp:37-49): three distributions, then the engine,
  // default-seeded; the constructor body draws t, b, then the feat normals of each bit (:28-37)
  std::normal_distribution<double> normal(0.0, sigma * sigma);  // sigma^2 as the std deviation, :22
  std::uniform_real_distribution<double> uniform_1(-1.0, 1.0);
  std::uniform_real_distribution<double> uniform_pi(0.0, 2.0 * M_PI);
  std::default_random_engine generator;
  for (uint32_t i = 0; i < bits; ++i) {
    t[i] = uniform_1(generator);
    b[i] = uniform_pi(generator);
    for (uint32_t j = 0; j < feat; ++j) w[(size_t)i * feat + j] = normal(generator);
  }
  return HS_OK;
}

hs_status hs_klsh_codes(int device, const uint8_t* classes, uint64_t n_residues,
                        const uint64_t* seq_start, uint64_t n_seq, const double* w, const double* b,
                        const double* t, uint32_t bits, uint64_t* codes, uint64_t* uncertain, char* err,
                        uint32_t err_cap) {
  auto say = [&](hs_status st, const std::string& msg) {
    if (err && err_cap) {
      strncpy(err, msg.c_str(), err_cap - 1);
      err[err_cap - 1] = 0;
    }
    return st;
  };
  if ((n_seq && (!seq_start || !codes)) || (n_residues && !classes) || !w || !b || !t || !bits || bits > 64)
    return say(HS_ERR_INVALID, "bad argument");
  if (!n_seq) return HS_OK;
  for (uint64_t s = 0; s < n_seq; ++s)
    if (seq_start[s] > seq_start[s + 1] || seq_start[s + 1] > n_residues)
      return say(HS_ERR_INVALID, "seq_start must be ascending and end within n_residues");
  for (uint64_t i = 0; i < n_residues; ++i)
    if (classes[i] >= HS_KLSH_CLASSES) return say(HS_ERR_INVALID, "residue class outside 0..7");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
    return say(HS_ERR_NO_DEVICE, "no usable gfx950 device");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return say(HS_ERR_NO_DEVICE, "no usable gfx950 device");
#define HS_KL(expr)                                                                          \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return say(e_ == hipErrorOutOfMemory ? HS_ERR_NOMEM : HS_ERR_HIP,                      \
                 std::string(#expr) + ": " + hipGetErrorString(e_));                         \
  } while (0)
  HS_KL(hipSetDevice(device));
  DevBuf d_cls, d_start, d_w, d_b, d_t, d_codes, d_unc;
  struct Guard {
    DevBuf* bufs[7];
    ~Guard() { for (DevBuf* x : bufs) x->release(); }
  } guard = {{&d_cls, &d_start, &d_w, &d_b, &d_t, &d_codes, &d_unc}};
  HS_KL(d_cls.reserve(std::max<size_t>(16, (size_t)n_residues)));
  HS_KL(d_start.reserve(((size_t)n_seq + 1) * 8));
  HS_KL(d_w.reserve((size_t)bits * HS_KLSH_FEATURES * 8));
  HS_KL(d_b.reserve((size_t)bits * 8));
  HS_KL(d_t.reserve((size_t)bits * 8));
  HS_KL(d_codes.reserve((size_t)n_seq * 8));
  HS_KL(d_unc.reserve((size_t)n_seq * 8));
  if (n_residues) HS_KL(hipMemcpy(d_cls.p, classes, (size_t)n_residues, hipMemcpyHostToDevice));
  HS_KL(hipMemcpy(d_start.p, seq_start, ((size_t)n_seq + 1) * 8, hipMemcpyHostToDevice));
  HS_KL(hipMemcpy(d_w.p, w, (size_t)bits * HS_KLSH_FEATURES * 8, hipMemcpyHostToDevice));
  HS_KL(hipMemcpy(d_b.p, b, (size_t)bits * 8, hipMemcpyHostToDevice));
  HS_KL(hipMemcpy(d_t.p, t, (size_t)bits * 8, hipMemcpyHostToDevice));
  HS_KL(hs_launch_klsh(d_cls.as<uint8_t>(), d_start.as<uint64_t>(), n_seq, d_w.as<double>(),
                       d_b.as<double>(), d_t.as<double>(), bits, d_codes.as<uint64_t>(),
                       d_unc.as<uint64_t>(), nullptr));
  HS_KL(hipDeviceSynchronize());
  HS_KL(hipMemcpy(codes, d_codes.p, (size_t)n_seq * 8, hipMemcpyDeviceToHost));
  if (uncertain) HS_KL(hipMemcpy(uncertain, d_unc.p, (size_t)n_seq * 8, hipMemcpyDeviceToHost));
#undef HS_KL
  return HS_OK;
}

// ---- persistent index --------------------------------------------------------------------------
// File = IndexFileHeader, then the payload: planes a, b, coordinate table (32 x 8 doubles), codes
// [n][k], and per table ids [n] u32, dir_key [nb] u64, dir_start [nb + 1] u32, dir_tuple [nb][K] i32.
// The header carries the payload's length and a 64-bit hash of it; hs_index_load checks both, then
// checks every table's CONTENT on the device before any kernel indexes with it (ids a permutation
// of 0..n-1 ascending inside a bucket, boundaries strictly ascending from 0 to n, fingerprints
// strictly ascending and equal to the fingerprint of the bucket's tuple): a corrupt, stale or
// hand-edited file gives HS_ERR_IO, never an out-of-bounds access.
namespace {
struct IndexFileHeader {
  char magic[8];  // "HSIDX002"
  uint32_t k, K, L, alphabet;
  double W;
  uint64_t n;
  uint32_t key_seed, pad;
  uint64_t n_buckets[HS_MAX_L], max_bucket[HS_MAX_L];
  uint64_t payload_bytes, payload_hash;
};
const char kIndexMagic[9] = "HSIDX002";
struct FileCloser {
  FILE* f;
  ~FileCloser() { if (f) fclose(f); }
};
// order-dependent 64-bit hash over the payload as a sequence of sections (same sequence on both sides)
struct PayloadHash {
  uint64_t h = 0x9e3779b97f4a7c15ull, bytes = 0;
  static uint64_t mix(uint64_t h, uint64_t w) {
    h = (h ^ w) * 0xff51afd7ed558ccdull;
    return h ^ (h >> 29);
  }
  void add(const void* p, size_t n) {
    const unsigned char* c = static_cast<const unsigned char*>(p);
    h = mix(h, (uint64_t)n);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
      uint64_t w;
      memcpy(&w, c + i, 8);
      h = mix(h, w);
    }
    if (i < n) {
      uint64_t w = 0;
      memcpy(&w, c + i, n - i);
      h = mix(h, w);
    }
    bytes += n;
  }
};
const size_t kFileChunk = (size_t)64 << 20;
bool header_shape_ok(const IndexFileHeader& hd) {
  if (memcmp(hd.magic, kIndexMagic, 8) != 0) return false;
  if (hd.k < 1 || hd.k > 75 || hd.K < 1 || hd.K > HS_MAX_K || hd.L < 1 || hd.L > HS_MAX_L ||
      hd.alphabet < 1 || hd.alphabet > HS_ALPHABET_PAD || hd.n >= (1ull << 31))
    return false;
  for (uint32_t l = 0; l < hd.L; ++l)
    if (hd.n_buckets[l] > hd.n || (hd.n != 0) != (hd.n_buckets[l] != 0)) return false;
  return true;
}
uint64_t payload_size(const IndexFileHeader& hd) {
  uint64_t b = (uint64_t)hd.L * hd.K * 8 * hd.k * 8 + (uint64_t)hd.L * hd.K * 8 + HS_ALPHABET_PAD * 8 * 8 + hd.n * hd.k;
  for (uint32_t l = 0; l < hd.L; ++l)
    b += hd.n * 4 + hd.n_buckets[l] * 8 + (hd.n_buckets[l] + 1) * 4 + hd.n_buckets[l] * hd.K * 4;
  return b;
}
}  // namespace

static hs_status write_device(hs_handle* h, FILE* f, const void* d_ptr, size_t bytes, std::vector<char>* tmp,
                              PayloadHash* ph) {
  tmp->resize(std::min(bytes, kFileChunk));
  for (size_t off = 0; off < bytes; off += kFileChunk) {
    const size_t m = std::min(kFileChunk, bytes - off);
    HS_HIP(h, hipMemcpy(tmp->data(), (const char*)d_ptr + off, m, hipMemcpyDeviceToHost));
    ph->add(tmp->data(), m);
    if (fwrite(tmp->data(), 1, m, f) != m) return fail(h, HS_ERR_IO, "short write to the index file");
  }
  return HS_OK;
}
static hs_status read_device(hs_handle* h, FILE* f, void* d_ptr, size_t bytes, std::vector<char>* tmp,
                             PayloadHash* ph) {
  tmp->resize(std::min(bytes, kFileChunk));
  for (size_t off = 0; off < bytes; off += kFileChunk) {
    const size_t m = std::min(kFileChunk, bytes - off);
    if (fread(tmp->data(), 1, m, f) != m) return fail(h, HS_ERR_IO, "index file truncated");
    ph->add(tmp->data(), m);
    HS_HIP(h, hipMemcpy((char*)d_ptr + off, tmp->data(), m, hipMemcpyHostToDevice));
  }
  return HS_OK;
}

hs_status hs_index_save(hs_handle* h, const char* path) {
  if (!h || !path) return HS_ERR_INVALID;
  if (!h->built) return fail(h, HS_ERR_STATE, "hs_index_build has not been called");
  HS_HIP(h, hipStreamSynchronize(h->stream));
  FILE* f = fopen(path, "wb");
  if (!f) return fail(h, HS_ERR_IO, std::string("cannot create ") + path);
  FileCloser closer = {f};
  IndexFileHeader hd;
  memset(&hd, 0, sizeof(hd));
  memcpy(hd.magic, kIndexMagic, 8);
  hd.k = h->p.k; hd.K = h->p.K; hd.L = h->p.L; hd.alphabet = (uint32_t)h->alphabet;
  hd.W = h->p.W; hd.n = h->n; hd.key_seed = h->key_seed;
  for (uint32_t l = 0; l < h->p.L; ++l) {
    hd.n_buckets[l] = h->info.n_buckets[l];
    hd.max_bucket[l] = h->info.max_bucket[l];
  }
  if (fwrite(&hd, sizeof(hd), 1, f) != 1) return fail(h, HS_ERR_IO, "short write to the index file");
  std::vector<char> tmp;
  PayloadHash ph;
  const size_t K = h->p.K, n = (size_t)h->n;
  HS_CHECK(write_device(h, f, h->a.p, (size_t)h->LK * h->d * 8, &tmp, &ph));
  HS_CHECK(write_device(h, f, h->b.p, (size_t)h->LK * 8, &tmp, &ph));
  HS_CHECK(write_device(h, f, h->coords.p, (size_t)HS_ALPHABET_PAD * 8 * 8, &tmp, &ph));
  HS_CHECK(write_device(h, f, h->codes.p, n * h->p.k, &tmp, &ph));
  for (uint32_t l = 0; l < h->p.L; ++l) {
    const size_t nb = (size_t)h->info.n_buckets[l];
    HS_CHECK(write_device(h, f, h->t_ids[l].p, n * 4, &tmp, &ph));
    HS_CHECK(write_device(h, f, h->t_dirkey[l].p, nb * 8, &tmp, &ph));
    HS_CHECK(write_device(h, f, h->t_dirstart[l].p, (nb + 1) * 4, &tmp, &ph));
    HS_CHECK(write_device(h, f, h->t_dirtuple[l].p, nb * K * 4, &tmp, &ph));
  }
  // the header again, now with the payload's length and hash
  hd.payload_bytes = ph.bytes;
  hd.payload_hash = ph.h;
  if (fseek(f, 0, SEEK_SET) != 0 || fwrite(&hd, sizeof(hd), 1, f) != 1 || fflush(f) != 0)
    return fail(h, HS_ERR_IO, "short write to the index file");
  return HS_OK;
}

hs_status hs_index_load(hs_handle* h, const char* path) {
  if (!h || !path) return HS_ERR_INVALID;
  hs_status st = ensure_device(h);
  if (st) return st;
  FILE* f = fopen(path, "rb");
  if (!f) return fail(h, HS_ERR_IO, std::string("cannot open ") + path);
  FileCloser closer = {f};
  IndexFileHeader hd;
  if (fread(&hd, sizeof(hd), 1, f) != 1 || !header_shape_ok(hd))
    return fail(h, HS_ERR_IO, "not an index file (or written by another version)");
  if (hd.k != h->p.k || hd.K != h->p.K || hd.L != h->p.L || hd.alphabet != (uint32_t)h->alphabet ||
      hd.W != h->p.W)
    return fail(h, HS_ERR_IO, "index file written for other parameters (k, K, L, W, alphabet)");
  if (hd.payload_bytes != payload_size(hd)) return fail(h, HS_ERR_IO, "index file inconsistent (payload length)");
  drop_index(h);
  memset(&h->prof, 0, sizeof(h->prof));
  memset(&h->info, 0, sizeof(h->info));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  std::vector<char> tmp, mine;
  PayloadHash ph;
  // planes and coordinate table must be the handle's, bit for bit
  const size_t sizes[3] = {(size_t)h->LK * h->d * 8, (size_t)h->LK * 8, (size_t)HS_ALPHABET_PAD * 8 * 8};
  const void* dev[3] = {h->a.p, h->b.p, h->coords.p};
  for (int i = 0; i < 3; ++i) {
    tmp.resize(sizes[i]);
    mine.resize(sizes[i]);
    if (fread(tmp.data(), 1, sizes[i], f) != sizes[i]) return fail(h, HS_ERR_IO, "index file truncated");
    ph.add(tmp.data(), sizes[i]);
    HS_HIP(h, hipMemcpy(mine.data(), dev[i], sizes[i], hipMemcpyDeviceToHost));
    if (memcmp(tmp.data(), mine.data(), sizes[i]) != 0)
      return fail(h, HS_ERR_IO, "index file written for other planes or another coordinate table");
  }
  const uint64_t n = hd.n;
  const int K = (int)h->p.K, L = (int)h->p.L, k = (int)h->p.k, PW = h->PW;
  h->n = n;
  h->key_seed = hd.key_seed;
  HS_HIP(h, h->codes.reserve(std::max<size_t>(16, (size_t)n * k)));
  HS_HIP(h, h->packed_all.reserve(std::max<size_t>(16, (size_t)n * PW * 16)));
  HS_HIP(h, h->counters.reserve(256));
  HS_CHECK(read_device(h, f, h->codes.p, (size_t)n * k, &tmp, &ph));
  if (n) {
    HS_HIP(h, hipMemsetAsync(h->counters.p, 0, 256, h->stream));
    HS_HIP(h, hs_launch_pack(h->codes.as<uint8_t>(), n, k, h->alphabet, h->packed_all.as<uint4>(),
                             h->counters.as<uint32_t>(), h->stream));
    uint32_t bad = 0;
    HS_HIP(h, hipMemcpyAsync(&bad, h->counters.p, 4, hipMemcpyDeviceToHost, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    if (bad) return fail(h, HS_ERR_IO, "index file holds a residue code outside the alphabet");
  }
  const bool with_rec8 = h->join8_tables_ok && k <= 50;
  HS_HIP(h, h->t_packed.reserve(((size_t)L * n + HS_JM_WAVE) * PW * 16));
  if (with_rec8) HS_HIP(h, h->t_rec8.reserve(((size_t)L * n + HS_JM_WAVE) * 16));
  const bool with_rho = with_rec8 && k <= 25 && !h->wide8;
  if (with_rho) HS_HIP(h, h->t_rho.reserve(((size_t)L * n + HS_JM_WAVE + 4) * 4));
  HS_HIP(h, h->t_pos.reserve(std::max<size_t>(16, (size_t)L * n * 4)));
  uint32_t* d_flag = h->counters.as<uint32_t>() + 16;  // [16] failure bits, [17 + l] largest bucket
  HS_HIP(h, hipMemsetAsync(d_flag, 0, (1 + HS_MAX_L) * 4, h->stream));
  for (int l = 0; l < L; ++l) {
    const size_t nb = (size_t)hd.n_buckets[l];
    uint4* const tab_packed = h->t_packed.as<uint4>() + (size_t)l * n * PW;
    HS_HIP(h, h->t_ids[l].reserve(std::max<size_t>(16, (size_t)n * 4)));
    HS_HIP(h, h->t_dirkey[l].reserve(std::max<size_t>(16, nb * 8)));
    HS_HIP(h, h->t_dirstart[l].reserve((nb + 1) * 4));
    HS_HIP(h, h->t_dirtuple[l].reserve(std::max<size_t>(16, nb * K * 4)));
    HS_CHECK(read_device(h, f, h->t_ids[l].p, (size_t)n * 4, &tmp, &ph));
    HS_CHECK(read_device(h, f, h->t_dirkey[l].p, nb * 8, &tmp, &ph));
    HS_CHECK(read_device(h, f, h->t_dirstart[l].p, (nb + 1) * 4, &tmp, &ph));
    HS_CHECK(read_device(h, f, h->t_dirtuple[l].p, nb * K * 4, &tmp, &ph));
    // content checks BEFORE anything indexes with the table; pos_of comes out of them
    HS_HIP(h, hs_launch_validate_table(h->t_ids[l].as<uint32_t>(), (uint32_t)n,
                                       h->t_pos.as<uint32_t>() + (size_t)l * n,
                                       h->t_dirstart[l].as<uint32_t>(), h->t_dirkey[l].as<uint64_t>(),
                                       h->t_dirtuple[l].as<int32_t>(), (uint32_t)nb, K, hd.key_seed, d_flag,
                                       d_flag + 1 + l, h->stream));
    uint32_t flag = 0;
    HS_HIP(h, hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    if (flag) {
      char msg[160];
      snprintf(msg, sizeof(msg), "index file corrupt: table %d fails its content checks (bits 0x%x: 1 id range, 2 id twice, "
               "4 boundaries, 8 key order, 16 key/tuple, 32 id order)", l, flag);
      return fail(h, HS_ERR_IO, msg);
    }
    if (with_rec8)
      HS_HIP(h, hs_launch_gather_rec8(h->packed_all.as<uint4>(), h->t_ids[l].as<uint32_t>(), (uint32_t)n,
                                      k, h->wide8, h->jtab8.p, h->jtab8.as<char>() + 1536,
                                      h->jtab8.as<float>() + 128, tab_packed,
                                      h->t_rec8.as<uint4>() + (size_t)l * n,
                                      with_rho ? h->t_rho.as<uint32_t>() + (size_t)l * n : nullptr, h->stream));
    else
      HS_HIP(h, hs_launch_gather_packed(h->packed_all.as<uint4>(), h->t_ids[l].as<uint32_t>(), n, PW,
                                        tab_packed, h->stream));
    hs_table_dev& tb = h->tabs.t[l];
    tb.dir_key = h->t_dirkey[l].as<uint64_t>();
    tb.dir_start = h->t_dirstart[l].as<uint32_t>();
    tb.dir_tuple = h->t_dirtuple[l].as<int32_t>();
    tb.packed = tab_packed;
    tb.ids = h->t_ids[l].as<uint32_t>();
    tb.pos_of = h->t_pos.as<uint32_t>() + (size_t)l * n;
    tb.nb = (uint32_t)nb;
    h->info.n_buckets[l] = nb;
  }
  uint32_t maxb[HS_MAX_L] = {0};
  HS_HIP(h, hipMemcpyAsync(maxb, d_flag + 1, (size_t)L * 4, hipMemcpyDeviceToHost, h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  for (int l = 0; l < L; ++l) h->info.max_bucket[l] = maxb[l];  // recomputed, not taken on trust
  char extra;
  if (ph.bytes != hd.payload_bytes || ph.h != hd.payload_hash || fread(&extra, 1, 1, f) != 0)
    return fail(h, HS_ERR_IO, "index file corrupt (payload hash or length does not match the header)");
  return finish_index(h);
}

// Host-only check of an index file (no GPU, no handle): header, payload length and hash, and the
// content rules hs_index_load enforces on the device.  HS_OK or HS_ERR_IO with a message.
hs_status hs_index_file_check(const char* path, char* err, uint32_t err_cap) {
  auto say = [&](hs_status st, const std::string& msg) {
    if (err && err_cap) {
      strncpy(err, msg.c_str(), err_cap - 1);
      err[err_cap - 1] = 0;
    }
    return st;
  };
  if (!path) return say(HS_ERR_INVALID, "null path");
  FILE* f = fopen(path, "rb");
  if (!f) return say(HS_ERR_IO, std::string("cannot open ") + path);
  FileCloser closer = {f};
  IndexFileHeader hd;
  if (fread(&hd, sizeof(hd), 1, f) != 1 || !header_shape_ok(hd))
    return say(HS_ERR_IO, "not an index file (or written by another version)");
  if (hd.payload_bytes != payload_size(hd)) return say(HS_ERR_IO, "payload length does not match the header");
  PayloadHash ph;
  std::vector<char> buf;
  auto section = [&](size_t bytes, std::vector<char>* keep) -> bool {
    std::vector<char>& dst = keep ? *keep : buf;
    if (keep) {
      dst.resize(bytes);
      if (bytes && fread(dst.data(), 1, bytes, f) != bytes) return false;
      // hashed in the chunks read_device/write_device use
      for (size_t off = 0; off < bytes; off += kFileChunk) ph.add(dst.data() + off, std::min(kFileChunk, bytes - off));
      return true;
    }
    dst.resize(std::min(bytes, kFileChunk));
    for (size_t off = 0; off < bytes; off += kFileChunk) {
      const size_t m = std::min(kFileChunk, bytes - off);
      if (fread(dst.data(), 1, m, f) != m) return false;
      ph.add(dst.data(), m);
    }
    return true;
  };
  const size_t d = 8 * (size_t)hd.k, LK = (size_t)hd.L * hd.K, n = (size_t)hd.n, K = hd.K;
  // planes, offsets and the coordinate table are hashed one section each (sizes well under a chunk
  // for any admissible parameters except the planes of very long k-mers, chunked like the rest)
  std::vector<char> codes;
  if (!section(LK * d * 8, nullptr) || !section(LK * 8, nullptr) || !section(HS_ALPHABET_PAD * 8 * 8, nullptr) ||
      !section(n * hd.k, &codes))
    return say(HS_ERR_IO, "index file truncated");
  for (size_t i = 0; i < codes.size(); ++i)
    if ((unsigned char)codes[i] >= hd.alphabet) return say(HS_ERR_IO, "residue code outside the alphabet");
  std::vector<char> ids_b, key_b, start_b, tup_b;
  std::vector<unsigned char> seen;
  for (uint32_t l = 0; l < hd.L; ++l) {
    const size_t nb = (size_t)hd.n_buckets[l];
    if (!section(n * 4, &ids_b) || !section(nb * 8, &key_b) || !section((nb + 1) * 4, &start_b) ||
        !section(nb * K * 4, &tup_b))
      return say(HS_ERR_IO, "index file truncated");
    const uint32_t* ids = reinterpret_cast<const uint32_t*>(ids_b.data());
    const uint64_t* key = reinterpret_cast<const uint64_t*>(key_b.data());
    const uint32_t* start = reinterpret_cast<const uint32_t*>(start_b.data());
    const int32_t* tup = reinterpret_cast<const int32_t*>(tup_b.data());
    const std::string where = " (table " + std::to_string(l) + ")";
    seen.assign(n, 0);
    for (size_t i = 0; i < n; ++i) {
      if (ids[i] >= n) return say(HS_ERR_IO, "id out of range" + where);
      if (seen[ids[i]]) return say(HS_ERR_IO, "id listed twice" + where);
      seen[ids[i]] = 1;
    }
    if ((nb ? start[0] : 0u) != 0u || start[nb] != n) return say(HS_ERR_IO, "bucket boundaries do not span 0..n" + where);
    for (size_t b = 0; b < nb; ++b) {
      if (!(start[b] < start[b + 1]) || start[b + 1] > n) return say(HS_ERR_IO, "bucket boundaries not ascending" + where);
      if (b + 1 < nb && !(key[b] < key[b + 1])) return say(HS_ERR_IO, "fingerprints not ascending" + where);
      if (hs_key_of(tup + b * K, (int)K, hd.key_seed) != key[b])
        return say(HS_ERR_IO, "a bucket's tuple does not have its fingerprint" + where);
      for (uint32_t i = start[b] + 1; i < start[b + 1]; ++i)
        if (!(ids[i - 1] < ids[i])) return say(HS_ERR_IO, "ids not ascending inside a bucket" + where);
    }
  }
  char extra;
  if (ph.bytes != hd.payload_bytes || ph.h != hd.payload_hash || fread(&extra, 1, 1, f) != 0)
    return say(HS_ERR_IO, "payload hash or length does not match the header");
  return HS_OK;
}

hs_status hs_index_info_get(const hs_handle* h, hs_index_info* out) {
  if (!h || !out) return HS_ERR_INVALID;
  if (!h->built) return HS_ERR_STATE;
  *out = h->info;
  return HS_OK;
}

// ------------------------------------------------------------------------------------- query
// Where a batch's ordered hits go (run_query's output arrays from its running total on) when the
// batch orders them itself (hs_launch_hit_order); ordered = true on return if it did.
struct BatchOut {
  uint32_t *q = nullptr, *id = nullptr, *table = nullptr;
  double* dist = nullptr;
  uint64_t room = 0;   // entries left in the arrays
  bool ordered = false;
};

// Counters block (h->counters): [0] prov_count u32, [1] hit_count u32, [2..3] cand_total u64, ...,
// [20] hits not ordered on the device, [21] HS_CNT_SURVIVOR_OVERFLOW, [32] the join's item counter.
// Internal status: the batch's filters passed more pairs than the 32-bit survivor counter holds.
static const hs_status HS_SPLIT_BATCH = (hs_status)1000;
// Wide int8 rows (all 8 coordinate columns) for this call?  Always for short k-mers (the index's
// member records are wide then).  For k = 21..25 when the radius is large for the k-mer length: the
// 4-column squared distance of two random k-mers is ~ N(k m, k v) (m, v: one residue pair), and once
// R^2 comes within 3 standard deviations of its mean the 4-column bound passes > 1e-3 of the bucket
// mates -- the survivor path, not the matrix pipe, then sets the pace (k = 25, R = 50: 1 % pass).
static bool want_wide(const hs_handle* h, double R) {
  if (h->wide8) return true;
  if (!h->wide8_ok || h->p.k > 25 || h->knobs.no_wide_by_radius) return false;
  if (h->knobs.force_wide) return true;
  const double k = (double)h->p.k, sd = sqrt(k * h->pair4_var);
  return R * R > k * h->pair4_mean - 3.0 * sd;
}

// The wide rows' member records for k = 21..25 (16 bytes per entry and table), built on first use
static hs_status ensure_rec8w(hs_handle* h) {
  if (h->wide8 || h->rec8w_ready) return HS_OK;
  const size_t n = h->n;
  const int L = (int)h->p.L;
  HS_HIP(h, h->t_rec8w.reserve(((size_t)L * n + HS_JM_WAVE) * 16));
  for (int l = 0; l < L; ++l)
    HS_HIP(h, hs_launch_gather_rec8(h->packed_all.as<uint4>(), h->tabs.t[l].ids, (uint32_t)n, (int)h->p.k, 1,
                                    h->jtab8.p, h->jtab8.as<char>() + 1536, h->jtab8.as<float>() + 128,
                                    nullptr, h->t_rec8w.as<uint4>() + (size_t)l * n, nullptr, h->stream));
  h->rec8w_ready = true;
  return HS_OK;
}

// May a self-join at radius R run from the residue codes alone (query_batch's self_codes)?  Only when
// nothing on its way can need the embedded centres: the int8 join and its thin-segment filter must
// apply, and no query row may be unrepresentable -- for a k-mer of the coordinate table the one way
// is -gamma overflowing its 13 base-127 digits, bounded here from R and the scale alone.
static bool self_codes_ok(const hs_handle* h, double R) {
  const bool wide = want_wide(h, R);
  const double r2 = R * R, s = wide ? h->join8_scale_w : h->join8_scale, k = (double)h->p.k;
  if (!h->join8_tables_ok || h->p.k > 50 || h->verify_mode == 1 || h->verify_mode == 3 || !(r2 < 30000.0))
    return false;
  // (segments routed away from the join go to the streaming filter, which works from the centres)
  if (h->join_min_q > 1 || h->join_min_m > 1 || h->knobs.no_self_codes) return false;
  // -gamma <= s^2 R^2 / 2 + L1(c^)/2 + 3, L1(c^) <= 127 * 4 k (127 * 8 k with wide rows)
  return s > 0.0 && 0.5 * s * s * r2 + (wide ? 508.0 : 254.0) * k + 3.0 < 127.0 * 127.0 * 13.0;
}

// d_qcodes_ext != null: the queries are k-mers given as residue codes [nq][k] (hs_query_codes) -- rows of
// the coordinate table like the DB's -- and d_centers is unused.
static hs_status query_batch(hs_handle* h, const double* d_centers, const uint8_t* d_qcodes_ext, uint32_t nq,
                             uint32_t q_base, double R, bool brute, uint64_t* d_cand, uint32_t* n_batch_hits,
                             BatchOut* bout = nullptr, bool allow_async = true) {
  const int K = (int)h->p.K, L = brute ? 1 : (int)h->p.L, k = (int)h->p.k;
  const uint32_t nql = nq * (uint32_t)L;
  const double r2 = R * R;  // motif_both_points.cpp:204
  const float r2_hi = filter_bound(r2);
  const int n_blocks = h->n_cu * 8;
  uint32_t* d_cnt = h->counters.as<uint32_t>();
  HS_HIP(h, hipMemsetAsync(d_cnt, 0, 256, h->stream));  // incl. the join's item counter (d_cnt + 32)
  HS_HIP(h, hipEventRecord(h->ev[0], h->stream));
  // fp16 form: k <= 25 only; int8 form (hs_join8.hip): k <= 50 (6 or 8 k-steps for two packed words)
  const bool can16 = h->join_tables_ok && k <= 25;
  const bool can8 = h->join8_tables_ok && k <= 50 && h->verify_mode != 3;
  bool use_join = !brute && h->verify_mode != 1 && r2 < 30000.0 && (can16 || can8);
  // int8 form of the join filter unless forced to fp16 (mode 3) or not representable
  bool use_i8 = use_join && can8;
  // survivors of the int8 join's 4-column bound pass an 8-column int8 bound before the exact
  // decision (hs_refine8_kernel); HS_NO_REFINE8 switches it off
  // (wide rows already hold all 8 columns: nothing to refine)
  const int wide = (use_i8 && want_wide(h, R)) ? 1 : 0;
  if (wide) HS_CHECK(ensure_rec8w(h));
  const uint4* const rec8 = (wide && !h->wide8) ? h->t_rec8w.as<uint4>() : h->t_rec8.as<uint4>();
  const void* const jtab_rows = wide ? (const void*)(h->jtab8.as<char>() + 1536) : (const void*)h->jtab8.p;
  const bool refine = use_i8 && !wide && !h->knobs.no_refine8;
  uint32_t* d_unsafe = d_cnt + 8;
  // Self-join (the queries are the indexed k-mers self_first + q_base ..): every per-query quantity
  // comes from the residue codes and the tables -- no embedded centres, no hashing, no directory
  // search (a k-mer probes the bucket it sits in).  Needs the int8 join with its thin-segment filter,
  // the only filters that work without per-query distance tables.
  // how the probes are grouped by bucket in front of the join: a counting sort over the bucket slots,
  // or -- when those far outnumber the probes -- a sort of the probes (HS_OPT_SEG_MODE forces one)
  // (measured at the configs[2] shape, 10^6 queries x 32 tables against 1.3e8 bucket slots -- a ratio of 4:
  // 11.3 ms for the whole probe + segment chain with the sort, 15.0 with the counting sort)
  bool seg_sparse = (uint64_t)h->nb_total > 2ull * nql;
  // bucket partition: 1/n_parts of the probes find their bucket; those are compacted before the grouping,
  // which then is the sort of the probes whatever the number of buckets
  const bool parted = h->bucket_parts > 1 && h->self_first == HS_NO_SELF;  // (searches only, not the self-joins)
  if (parted) seg_sparse = true;
  uint32_t nqs = nql;  // probes the grouping works on (bucket partition: the ones that found a bucket)
  const uint32_t* owned_list = nullptr;  // bucket partition: the probes of this part, ascending (device)
  uint32_t n_owned = 0;
  if (h->knobs.seg_mode) seg_sparse = h->knobs.seg_mode == 1;
  const bool self_codes = h->self_first != HS_NO_SELF && !brute && use_i8 && self_codes_ok(h, R);
  const uint8_t* d_qcodes =
      self_codes ? h->codes.as<uint8_t>() + ((uint64_t)h->self_first + q_base) * k : nullptr;
  // Queries given as codes (hs_query_codes): a checked copy first (a code outside the alphabet is
  // reported with the batch's counters and replaced by 0, so no kernel indexes a table with it).  When
  // every filter on the way works from codes (self_codes_ok: the int8 join and its thin-segment filter)
  // no embedded centre exists at any point: 25 bytes per query instead of 1600 -- hash, query rows and
  // the exact decision all read the table rows the DB's k-mers read.  Otherwise the codes are embedded
  // here, on the device, and the batch runs as for any other centres.
  bool ext_codes = false;
  if (d_qcodes_ext) {
    HS_HIP(h, h->qcodes_buf.reserve(std::max<size_t>(16, (size_t)nq * k)));
    HS_HIP(h, hs_launch_check_codes(d_qcodes_ext, (uint64_t)nq * k, h->alphabet, h->qcodes_buf.as<uint8_t>(),
                                    d_cnt + HS_CNT_BAD_QUERY_CODE, h->stream));
    if (!brute && use_i8 && self_codes_ok(h, R)) {
      ext_codes = true;
      d_qcodes = h->qcodes_buf.as<uint8_t>();
    } else {
      HS_HIP(h, h->qembed.reserve(std::max<size_t>(16, (size_t)nq * h->d * 8)));
      HS_HIP(h, hs_launch_embed(h->qcodes_buf.as<uint8_t>(), nq, k, h->coords.as<double>(), h->qembed.as<double>(),
                                h->stream));
      d_centers = h->qembed.as<double>();
    }
  }
  const bool from_codes = self_codes || ext_codes;  // no centres: every per-query quantity from the codes
  if (use_join) {
    // the join filter's query rows depend on the centres only: quantised on the side stream while
    // the main stream hashes and probes (both passes stream the same 8d bytes per query)
    HS_HIP(h, h->c16.reserve((size_t)nq * 208 * 2));
    if (refine) HS_HIP(h, h->c8b.reserve((size_t)nq * hs_join8_row_bytes(k, wide)));
    HS_HIP(h, hipEventRecord(h->evx[EV_FORK], h->stream));
    HS_HIP(h, hipStreamWaitEvent(h->stream2, h->evx[EV_FORK], 0));
    if (from_codes)
      HS_HIP(h, hs_launch_qprep8_codes(d_qcodes, nq, k, wide, r2, h->coords.as<double>(), h->jtab8.p,
                                       h->jtab8.as<char>() + 1024, h->jtab8.as<char>() + 1536,
                                       h->jtab8.as<float>() + 128, h->c16.p, refine ? h->c8b.p : nullptr,
                                       h->stream2));
    else if (use_i8)
      HS_HIP(h, hs_launch_qprep8(d_centers, nq, k, wide, r2, h->jtab8.as<float>() + 128, h->c16.p, d_unsafe,
                                 refine ? h->c8b.p : nullptr, h->stream2));
    else
      HS_HIP(h, hs_launch_qprep(d_centers, nq, k, r2, h->c16.p, d_unsafe, h->stream2));
    HS_HIP(h, hipEventRecord(h->evx[EV_JOIN], h->stream2));
  }
  if (!brute) {
    HS_HIP(h, h->qints.reserve((size_t)nq * h->LK * 4));
    HS_HIP(h, h->qstart.reserve(((size_t)nql + HS_QRANGE_PAD) * 4));
    HS_HIP(h, h->qcount.reserve(((size_t)nql + HS_QRANGE_PAD) * 4));
    HS_HIP(h, h->nslices.reserve(((size_t)nql + 1) * 4));
    HS_HIP(h, h->probe_slow.reserve(((size_t)nql + 1) * 4));
    HS_HIP(h, h->slice_off.reserve(((size_t)nql + 1) * 4));
    HS_HIP(h, h->temp.reserve(hs_scan_u32_temp((size_t)nql + 1) + 256));
    if (ext_codes)
      HS_CHECK(hash_dispatch(h, d_qcodes, nullptr, nq, -1, h->qints.as<int32_t>(), h->LK, 2, h->stream));
    else if (!self_codes)
      HS_CHECK(hash_dispatch(h, nullptr, d_centers, nq, -1, h->qints.as<int32_t>(), h->LK, 2, h->stream));
  }
  HS_HIP(h, hipEventRecord(h->ev[1], h->stream));
  // Bucket join when fp16 / int8 can carry the data (decided above); the streaming kernel otherwise
  // (and for brute force).  Both append survivors to one list in front of the same exact decision.
  if (!brute) {
    // with a join ahead, the probe also numbers each probe's bucket and ranks it inside (the
    // grouping of the probes by bucket is then a counting sort: hs_launch_seg_group)
    uint32_t *bucket_count = nullptr, *qbucket = nullptr, *qrank = nullptr;
    if (use_join) {
      HS_HIP(h, h->seg_keys.reserve(((size_t)nql + 1) * 8));
      qbucket = h->seg_keys.as<uint32_t>();
      if (seg_sparse) {
        // buckets far outnumber probes: the probes are sorted on their bucket number instead (no
        // ranks, no pass over the bucket slots: hs_launch_seg_group_sparse)
        HS_HIP(h, h->bucket_work.reserve(4 * ((size_t)nql + 1) * 4));
      } else {
        HS_HIP(h, h->bucket_work.reserve(4 * ((size_t)h->nb_total + 2) * 4));
        bucket_count = h->bucket_work.as<uint32_t>();
        qrank = qbucket + ((size_t)nql + 1);
      }
    }
    HS_HIP(h, hs_launch_set_u32(h->nslices.as<uint32_t>() + nql, 0u, h->stream));
    if (self_codes)
      HS_HIP(h, hs_launch_self_probe(h->tabs, h->self_first + q_base, nq, L, h->qstart.as<uint32_t>(),
                                     h->qcount.as<uint32_t>(), h->nslices.as<uint32_t>(), d_cand,
                                     reinterpret_cast<unsigned long long*>(d_cnt + 2),
                                     h->dir_base.as<uint32_t>(), h->nb_total, bucket_count, qbucket, qrank,
                                     h->stream));
    else {
      // bucket partition with a join ahead: the part's own probes (by their bucket ints alone) are listed first,
      // and only those -- 1 / n_parts of the batch -- pay for a fingerprint and a walk of the directory
      if (parted && use_join) {
        const size_t n1 = (size_t)nql + 1;
        HS_HIP(h, h->part_work.reserve(5 * n1 * 4));
        uint32_t* const pw = h->part_work.as<uint32_t>();
        HS_HIP(h, hs_launch_part_owned(probe_tabs(h, q_base), h->qints.as<int32_t>(), nq, K, L, h->nb_total, pw,
                                       h->qstart.as<uint32_t>(), h->qcount.as<uint32_t>(), h->nslices.as<uint32_t>(),
                                       d_cand, qbucket, h->stream));
        HS_HIP(h, hs_exclusive_scan_u32(h->temp.p, h->temp.cap, pw, pw + n1, n1, h->stream));
        HS_HIP(h, hs_launch_flagged_list(pw, pw + n1, nql, pw + 4 * n1, h->stream));
        HS_HIP(h, hipMemcpyAsync(&n_owned, pw + n1 + nql, 4, hipMemcpyDeviceToHost, h->stream));
        HS_HIP(h, hipStreamSynchronize(h->stream));
        owned_list = pw + 4 * n1;
      }
      HS_HIP(h, hs_launch_probe(probe_tabs(h, q_base, owned_list, n_owned), h->qints.as<int32_t>(), nq, K, L, h->key_seed,
                                h->qstart.as<uint32_t>(), h->qcount.as<uint32_t>(),
                                h->nslices.as<uint32_t>(), d_cand,
                                reinterpret_cast<unsigned long long*>(d_cnt + 2),
                                h->probe_slow.as<uint32_t>(), h->dir_base.as<uint32_t>(), h->nb_total,
                                bucket_count, qbucket, qrank, h->stream));
    }
  }
  unsigned long long* d_jstats = reinterpret_cast<unsigned long long*>(d_cnt + 10);
  uint32_t n_items = 0, n_slices = 1, jm = HS_JM_BLOCK;
  bool async_items = false, use_r = false;
  const int seg_shift = seg_shift_of(h);
  if (use_join) {
    const size_t n1 = (size_t)nql + 1;
    HS_HIP(h, h->c16s.reserve(((size_t)nql + 64) * 208 * 2));
    HS_HIP(h, h->seg_keys.reserve(n1 * 8));
    HS_HIP(h, h->seg_keys_sorted.reserve(n1 * 8));
    HS_HIP(h, h->seg_vals.reserve(n1 * 4));
    HS_HIP(h, h->sorted_ql.reserve(n1 * 4));
    HS_HIP(h, h->seg_key.reserve(n1 * 8));
    HS_HIP(h, h->seg_cnt.reserve(n1 * 4));
    HS_HIP(h, h->seg_qoff.reserve(n1 * 4));
    HS_HIP(h, h->seg_items.reserve(n1 * 4));
    HS_HIP(h, h->item_off.reserve(n1 * 4));
    HS_HIP(h, h->seg_n.reserve(64));
    HS_HIP(h, h->seg_of.reserve(n1 * 4));
    HS_HIP(h, h->temp.reserve(std::max(hs_scan_u32_temp(n1), hs_scan_u32_temp((size_t)h->nb_total + 2)) + 256));
    // (the query rows of the join filter were quantised on the side stream, beside hash and probe)
    HS_HIP(h, hipStreamWaitEvent(h->stream, h->evx[EV_JOIN], 0));
    HS_HIP(h, hipMemsetAsync(h->seg_cnt.p, 0, n1 * 4, h->stream));
    if (seg_sparse) {
      HS_HIP(h, h->temp.reserve(std::max(hs_sort_pairs_u32_u32_temp(nql), hs_scan_u32_temp(n1)) + 256));
      const uint32_t* keys_in = h->seg_keys.as<uint32_t>();
      const uint32_t* probes_in = nullptr;
      if (parted) {
        HS_HIP(h, h->part_work.reserve(5 * n1 * 4));
        uint32_t* const pw = h->part_work.as<uint32_t>();
        const uint32_t n_cand = owned_list ? n_owned : nql;  // (the part's own probes, listed ahead of the probe kernel)
        uint32_t n_found = 0;
        if (n_cand) {
          HS_HIP(h, hs_launch_found_probes(h->seg_keys.as<uint32_t>(), n_cand, h->nb_total, h->temp.p, h->temp.cap, pw,
                                           pw + n1, pw + 2 * n1, pw + 3 * n1, h->stream, owned_list));
          HS_HIP(h, hipMemcpyAsync(&n_found, pw + n1 + n_cand, 4, hipMemcpyDeviceToHost, h->stream));
        }
        HS_HIP(h, hipStreamSynchronize(h->stream));
        if (n_found) {  // (none at all: the batch goes on as one of probes that found nothing)
          nqs = n_found;
          keys_in = pw + 2 * n1;
          probes_in = pw + 3 * n1;
        }
      }
      HS_HIP(h, hs_launch_seg_group_sparse(h->tabs, h->dir_base.as<uint32_t>(), L, seg_shift, h->nb_total,
                                           h->temp.p, h->temp.cap, keys_in,
                                           h->seg_keys.as<uint32_t>() + n1, h->seg_vals.as<uint32_t>(),
                                           h->bucket_work.as<uint32_t>(), nqs, h->sorted_ql.as<uint32_t>(),
                                           h->seg_key.as<uint64_t>(), h->seg_cnt.as<uint32_t>(),
                                           h->seg_n.as<uint32_t>(), h->seg_of.as<uint32_t>(), h->stream, probes_in));
    } else
    HS_HIP(h, hs_launch_seg_group(h->tabs, h->dir_base.as<uint32_t>(), L, seg_shift, h->nb_total,
                                  h->bucket_work.as<uint32_t>(),
                                  h->bucket_work.as<uint32_t>() + ((size_t)h->nb_total + 2), h->temp.p,
                                  h->temp.cap, h->seg_keys.as<uint32_t>(),
                                  h->seg_keys.as<uint32_t>() + n1, nql, h->sorted_ql.as<uint32_t>(),
                                  h->seg_key.as<uint64_t>(), h->seg_cnt.as<uint32_t>(),
                                  h->seg_n.as<uint32_t>(), h->seg_of.as<uint32_t>(), h->stream));
    HS_HIP(h, hs_exclusive_scan_u32(h->temp.p, h->temp.cap, h->seg_cnt.as<uint32_t>(),
                                    h->seg_qoff.as<uint32_t>(), (size_t)nqs + 1, h->stream));
    // work items: one wave's 128 members for the wave-independent int8 join, 512 otherwise
    jm = use_i8 ? hs_join8_members_per_item(k, wide) : HS_JM_BLOCK;
    // k <= 25 with 4-column rows: segments probed by at most HS_JR_MAXQ queries of the batch go to the
    // query-resident kernel (hs_join8r_kernel), as the tail of the item list
    if (h->resident_nq && (nq > 2 * h->resident_nq || 2 * nq < h->resident_nq)) h->resident_share = -1.0;
    use_r = use_i8 && !wide && k <= 25 && h->alphabet <= HS_JR_MAX_ALPHABET && !h->knobs.no_join_r &&
            (h->knobs.force_join_r || h->resident_share < 0.0 || h->resident_share >= 0.5 || h->resident_age >= 64);
    HS_CHECK(cut_items(h, nqs, jm, d_jstats, use_r ? HS_JR_MAXQ : 0u, nql));
    if (use_i8)
      HS_HIP(h, hs_launch_gather_c8t(h->c16.p, h->sorted_ql.as<uint32_t>(), h->seg_qoff.as<uint32_t>(),
                                     h->seg_of.as<uint32_t>(), nqs, L, k, wide, h->c16s.p, h->stream));
    else
      HS_HIP(h, hs_launch_gather_c16(h->c16.p, h->sorted_ql.as<uint32_t>(), nqs, L, h->c16s.p, h->stream));
  }
  if (!brute) {
    // (every segment joined -- the default --: all slice counts are zero by now, and so is their scan)
    if (use_join && h->join_min_q == 1 && h->join_min_m == 1)
      HS_HIP(h, hipMemsetAsync(h->slice_off.p, 0, ((size_t)nql + 1) * 4, h->stream));
    else
      HS_HIP(h, hs_exclusive_scan_u32(h->temp.p, h->temp.cap, h->nslices.as<uint32_t>(),
                                      h->slice_off.as<uint32_t>(), (size_t)nql + 1, h->stream));
    // No host round trip when the int8 join takes every segment and the previous batch left a
    // capacity hint: the item count stays on the device (item_off[nql]); join legality and the
    // capacity are checked with the batch's final read-back, a violation repeats the batch the
    // slow way.  Otherwise one round trip: join legality, item count, streaming slices.
    async_items = allow_async && use_i8 && h->join_min_q == 1 && h->join_min_m == 1 && h->item_cap_hint &&
                  !h->knobs.sync_items;
    uint32_t unsafe = 0;
    if (async_items) {
      n_items = h->item_cap_hint;
      n_slices = 0;
    } else if (use_join) {
      HS_HIP(h, hipMemcpyAsync(&unsafe, d_unsafe, 4, hipMemcpyDeviceToHost, h->stream));
      HS_HIP(h, hipMemcpyAsync(&n_items, h->item_off.as<uint32_t>() + nqs, 4, hipMemcpyDeviceToHost,
                               h->stream));
    }
    if (!async_items) {
      HS_HIP(h, hipMemcpyAsync(&n_slices, h->slice_off.as<uint32_t>() + nql, 4, hipMemcpyDeviceToHost,
                               h->stream));
      HS_HIP(h, hipStreamSynchronize(h->stream));
    }
    if (from_codes && unsafe) return fail(h, HS_ERR_STATE, "queries from codes: a query row marked unsafe");
    if (use_i8 && unsafe && !can16) {
      // a query int8 cannot carry and no fp16 form for this k: the batch streams (below)
      use_i8 = false;
    } else if (use_i8 && unsafe) {
      // a query int8 cannot carry: redo the query rows in fp16 (segments and items are shared)
      use_i8 = false;
      HS_HIP(h, hipMemsetAsync(d_unsafe, 0, 4, h->stream));
      HS_HIP(h, hs_launch_qprep(d_centers, nq, k, r2, h->c16.p, d_unsafe, h->stream));
      HS_HIP(h, hs_launch_gather_c16(h->c16.p, h->sorted_ql.as<uint32_t>(), nqs, L, h->c16s.p, h->stream));
      HS_HIP(h, hipMemcpyAsync(&unsafe, d_unsafe, 4, hipMemcpyDeviceToHost, h->stream));
      if (jm != HS_JM_BLOCK) {  // the fp16 kernel works on 512-member items: cut the segments again
        jm = HS_JM_BLOCK;
        use_r = false;
        HS_HIP(h, hipMemsetAsync(d_jstats, 0, 16, h->stream));
        HS_CHECK(cut_items(h, nqs, jm, d_jstats, 0u, nql));
        HS_HIP(h, hipMemcpyAsync(&n_items, h->item_off.as<uint32_t>() + nqs, 4, hipMemcpyDeviceToHost,
                                 h->stream));
      }
      HS_HIP(h, hipStreamSynchronize(h->stream));
    }
    if (use_join && unsafe) {
      // a query fp16 cannot carry: this batch streams entirely (re-derive the slice counts)
      use_join = false;
      n_items = 0;
      HS_HIP(h, hipMemsetAsync(d_cnt + 2, 0, 8, h->stream));
      HS_HIP(h, hs_launch_probe(probe_tabs(h, q_base), h->qints.as<int32_t>(), nq, K, L, h->key_seed,
                                h->qstart.as<uint32_t>(), h->qcount.as<uint32_t>(),
                                h->nslices.as<uint32_t>(), d_cand,
                                reinterpret_cast<unsigned long long*>(d_cnt + 2),
                                h->probe_slow.as<uint32_t>(), nullptr, 0, nullptr, nullptr, nullptr,
                                h->stream));
      HS_HIP(h, hs_exclusive_scan_u32(h->temp.p, h->temp.cap, h->nslices.as<uint32_t>(),
                                      h->slice_off.as<uint32_t>(), (size_t)nql + 1, h->stream));
      n_slices = 1;
    }
    if (n_items) {
      HS_HIP(h, h->item_desc.reserve((size_t)n_items * 32));
      HS_HIP(h, hs_launch_item_desc(h->tabs, h->seg_key.as<uint64_t>(), h->seg_cnt.as<uint32_t>(),
                                    h->seg_qoff.as<uint32_t>(), h->item_off.as<uint32_t>(), nqs,
                                    h->sorted_ql.as<uint32_t>(), h->qcount.as<uint32_t>(), n_items, jm,
                                    seg_shift, h->seg_vals.as<uint32_t>(), h->PW,
                                    async_items ? h->item_off.as<uint32_t>() + nqs : nullptr,
                                    h->item_desc.as<uint4>(), h->stream));
    }
  }
  // segments routed away from the join (HS_OPT_JOIN_MIN_Q / _M; none by default) go through the streaming
  // filter and its per-query distance tables, on the side stream beside the join
  const bool side = !brute && n_items && n_slices;
  if (brute || n_slices) {
    HS_HIP(h, h->tq.reserve((size_t)nq * k * HS_TROW * 4));
    if (!side)
      HS_HIP(h, hs_launch_qtables(d_centers, nq, k, h->coords.as<double>(), h->alphabet,
                                  h->tq.as<float>(), h->stream));
  }
  HS_HIP(h, hipEventRecord(h->ev[2], h->stream));
  bool tables_done = !side;
  uint32_t prov_cap = (uint32_t)std::max<size_t>(h->prov.cap / 8, std::max<size_t>(1u << 20, 16ull * nq));
  if (!h->pin_cnt) {
    HS_HIP(h, hipHostMalloc(reinterpret_cast<void**>(&h->pin_cnt), 64 * 4, hipHostMallocDefault));
    memset(h->pin_cnt, 0, 64 * 4);
  }
  // [0] survivors [1] hits [2..3] candidates ... [10..13] join statistics [20] order fallback
  uint32_t* const host_cnt = h->pin_cnt;
  uint32_t* const host_proj = h->pin_cnt + 32;  // MFMA projection of the queries: {slots reserved, values flagged}
  uint32_t& n_items_real = h->pin_cnt[40];
  memset(h->pin_cnt, 0, 64 * 4);
  const bool proj_stats = !brute && !self_codes && use_projection(h);
  double ms_verify = 0, ms_final = 0, ms_join = 0;
  uint32_t launches = 0;
  for (;;) {  // retried only when a workspace capacity was exceeded
    HS_HIP(h, h->prov.reserve((size_t)prov_cap * 8));
    uint32_t hit_cap = (uint32_t)std::max<size_t>(h->hit_key.cap / 8, prov_cap);
    HS_HIP(h, h->hit_key.reserve((size_t)hit_cap * 8));
    HS_HIP(h, h->hit_val.reserve((size_t)hit_cap * 8));
    HS_HIP(h, hipMemsetAsync(d_cnt, 0, 8, h->stream));
    // the batch orders its hits itself (bucket by query, no sort, no host count) unless brute force
    // (not tried again at a radius at which the previous batch had a query with too many hits for it:
    // the attempt costs 10 % of such a batch -- k = 15 at the C2 sizes, 545 hits per query)
    const bool order_here = bout && !brute && !h->knobs.sort_hits && !(h->order_failed && h->order_failed_R == R);
    uint32_t *qcnt = nullptr, *qoff = nullptr, *qfill = nullptr;
    if (order_here) {
      const size_t n1q = (size_t)nq + 1;
      HS_HIP(h, h->qhits.reserve((6 * n1q + 8) * 4));
      qcnt = h->qhits.as<uint32_t>();
      qoff = qcnt + n1q;
      qfill = qoff + n1q;  // (and behind it the lists of the queries a block orders: 8 + 3 nq words)
      HS_HIP(h, hipMemsetAsync(qcnt, 0, (3 * n1q + 8) * 4, h->stream));
      if (launches) HS_HIP(h, hipMemsetAsync(d_cnt + 20, 0, 4, h->stream));  // retry: the "too many hits" flag
      HS_HIP(h, h->hit_kv.reserve((size_t)hit_cap * 16));
      HS_HIP(h, h->hit_rank.reserve((size_t)hit_cap * 4));
      HS_HIP(h, h->temp.reserve(hs_scan_u32_temp(n1q) + 256));
    }
    HS_HIP(h, hipEventRecord(h->ev[3], h->stream));
    if (launches) HS_HIP(h, hipMemsetAsync(d_cnt + 32, 0, 64, h->stream));  // retry: the item counters again
    if (side) {
      HS_HIP(h, hipEventRecord(h->evx[EV_FORK], h->stream));
      HS_HIP(h, hipStreamWaitEvent(h->stream2, h->evx[EV_FORK], 0));
      if (!tables_done)
        HS_HIP(h, hs_launch_qtables(d_centers, nq, k, h->coords.as<double>(), h->alphabet,
                                    h->tq.as<float>(), h->stream2));
      tables_done = true;
      HS_HIP(h, hs_launch_verify(h->tabs, h->qstart.as<uint32_t>(), h->qcount.as<uint32_t>(),
                                 h->slice_off.as<uint32_t>(), nql, h->tq.as<float>(), k, L, r2_hi,
                                 d_cnt, prov_cap, h->prov.as<uint2>(), n_blocks, h->stream2));
      HS_HIP(h, hipEventRecord(h->evx[EV_JOIN], h->stream2));
    }
    if (brute) {
      HS_HIP(h, hs_launch_bruteforce(h->packed_all.as<uint4>(), (uint32_t)h->n, h->tq.as<float>(),
                                     nq, k, r2_hi, d_cnt, prov_cap, h->prov.as<uint2>(), nullptr,
                                     nullptr, n_blocks, h->stream));
    } else {
      HS_HIP(h, hipEventRecord(h->ev[11], h->stream));  // the join kernel alone: ev[11] .. ev[10]
      if (n_items && use_i8) {
        // the head [0, split) of the item list through the query-streaming kernel, the tail through the
        // query-resident one (split == the item count when no segment qualifies or use_r is off)
        const uint32_t* const d_split = h->seg_n.as<uint32_t>() + 2;
        // XCD-local runs of items where the batch's query tiles do not stay in every XCD's L2 anyway
        // (d_cnt + 40 .. 47: the per-XCD chunk counters)
        const uint64_t tile_bytes = (uint64_t)nqs * (uint64_t)hs_join8_row_bytes(k, wide);
        const uint32_t xcd_run = h->knobs.join_xcd_run >= 0 ? (uint32_t)h->knobs.join_xcd_run
                                                            : (tile_bytes > (64ull << 20) ? 128u : 0u);
        HS_HIP(h, hs_launch_join8w(h->item_desc.as<uint4>(), n_items, h->tabs.t[0].packed,
                                   rec8, h->c16s.p, jtab_rows, k, wide, d_cnt, prov_cap,
                                   h->prov.as<uint2>(), d_cnt + 32, h->n_cu * h->join_blocks_per_cu,
                                   use_r ? d_split : (async_items ? h->item_off.as<uint32_t>() + nqs : nullptr),
                                   h->pairs_per_item, xcd_run, h->stream, h->knobs.join_chunk));
        if (use_r)
          HS_HIP(h, hs_launch_join8r(h->item_desc.as<uint4>(), n_items, d_split, h->tabs.t[0].packed,
                                     h->t_rho.as<uint32_t>(),
                                     h->c16s.p, jtab_rows, h->alphabet, d_cnt, prov_cap, h->prov.as<uint2>(),
                                     d_cnt + 33, h->n_cu * h->join_blocks_per_cu, h->pairs_per_item, h->stream));
      }
      else if (n_items)
        HS_HIP(h, hs_launch_join(h->item_desc.as<uint4>(), n_items, h->tabs.t[0].packed,
                                 h->sorted_ql.as<uint32_t>(),
                                 h->c16s.p, h->jtab.p,
                                 reinterpret_cast<const float*>(h->jtab.as<char>() + 512), k, d_cnt,
                                 prov_cap, h->prov.as<uint2>(), h->n_cu * h->join_blocks_per_cu, h->stream));
      HS_HIP(h, hipEventRecord(h->ev[10], h->stream));
      if (side)
        HS_HIP(h, hipStreamWaitEvent(h->stream, h->evx[EV_JOIN], 0));
      else if (n_slices)
        HS_HIP(h, hs_launch_verify(h->tabs, h->qstart.as<uint32_t>(), h->qcount.as<uint32_t>(),
                                   h->slice_off.as<uint32_t>(), nql, h->tq.as<float>(), k, L, r2_hi,
                                   d_cnt, prov_cap, h->prov.as<uint2>(), n_blocks, h->stream));
    }
    HS_HIP(h, hipEventRecord(h->ev[4], h->stream));
    if (brute) {
      HS_HIP(h, hs_launch_bf_finalize(h->codes.as<uint8_t>(), d_centers, h->coords.as<double>(),
                                      h->prov.as<uint2>(), d_cnt, prov_cap, k, R, q_base, d_cnt + 1,
                                      hit_cap, h->hit_key.as<uint64_t>(), h->hit_val.as<uint64_t>(),
                                      h->stream));
    } else {
      const uint2* fin_list = h->prov.as<uint2>();
      const uint32_t* fin_count = d_cnt;
      if (refine && use_i8 && n_items) {
        // use_i8 may have been dropped after the readback (fp16 fallback): then no second row exists
        HS_HIP(h, h->prov2.reserve((size_t)prov_cap * 8));
        HS_HIP(h, hipMemsetAsync(d_cnt + 4, 0, 4, h->stream));
        HS_HIP(h, hs_launch_refine8(h->tabs, h->prov.as<uint2>(), d_cnt, prov_cap,
                                    h->sorted_ql.as<uint32_t>(), h->c16.p, h->c8b.p,
                                    h->jtab8.as<char>() + 1024, h->jtab8.as<float>() + 128, k, L,
                                    h->qstart.as<uint32_t>(), h->qcount.as<uint32_t>(),
                                    h->prov2.as<uint2>(), d_cnt + 4, h->stream));
        fin_list = h->prov2.as<uint2>();
        fin_count = d_cnt + 4;
      }
      const uint4* d_qpacked = nullptr;
      if (d_qcodes && k <= 75) {  // the queries are k-mers: packed like the members, for the exact pass
        HS_HIP(h, h->qpacked.reserve(std::max<size_t>(16, (size_t)nq * h->PW * 16)));
        HS_HIP(h, hs_launch_pack(d_qcodes, nq, k, h->alphabet, h->qpacked.as<uint4>(),
                                 d_cnt + HS_CNT_BAD_QUERY_CODE, h->stream));
        d_qpacked = h->qpacked.as<uint4>();
      }
      HS_HIP(h, hs_launch_finalize(h->tabs, h->codes.as<uint8_t>(), d_centers, d_qcodes, h->coords.as<double>(),
                                   h->qstart.as<uint32_t>(), h->qcount.as<uint32_t>(),
                                   fin_list, fin_count, prov_cap, h->sorted_ql.as<uint32_t>(),
                                   k, L, r2, h->sqrt_test ? R : (double)NAN, q_base, h->self_first, d_cnt + 1,
                                   hit_cap, h->hit_key.as<uint64_t>(), h->hit_val.as<uint64_t>(), qcnt,
                                   h->alphabet, d_qpacked, order_here ? h->hit_rank.as<uint32_t>() : nullptr,
                                   h->stream));
      if (order_here) {
        const size_t n1q_ = (size_t)nq + 1;
        HS_HIP(h, hs_exclusive_scan_u32(h->temp.p, h->temp.cap, qcnt, qoff, (size_t)nq + 1, h->stream));
        HS_HIP(h, hs_launch_hit_order(h->hit_key.as<uint64_t>(), h->hit_val.as<uint64_t>(), d_cnt + 1, hit_cap,
                                      q_base, nq, qoff, h->hit_rank.as<uint32_t>(), h->hit_kv.p, d_cnt + 20, qfill + n1q_, bout->q, bout->id,
                                      bout->table, bout->dist, bout->room, h->n_cu, h->stream));
      }
    }
    HS_HIP(h, hipEventRecord(h->ev[5], h->stream));
    HS_HIP(h, hipMemcpyAsync(host_cnt, d_cnt, 96, hipMemcpyDeviceToHost, h->stream));
    if (proj_stats)
      HS_HIP(h, hipMemcpyAsync(host_proj, h->proj_cnt.as<uint32_t>() + 4, 8, hipMemcpyDeviceToHost, h->stream));
    if (async_items)
      HS_HIP(h, hipMemcpyAsync(&n_items_real, h->item_off.as<uint32_t>() + nqs, 4, hipMemcpyDeviceToHost,
                               h->stream));
    if (use_join && use_i8 && n_items)  // [41] first item of the few-query class, [42] items (cut_items)
      HS_HIP(h, hipMemcpyAsync(h->pin_cnt + 41, h->seg_n.as<uint32_t>() + 2, 8, hipMemcpyDeviceToHost, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    // > ~4e9 survivors: run_query halves the batch (HS_TEST_SPLIT_ABOVE=n: as if every batch of more
    // than n queries had overflowed -- the tests' handle on the splitting logic)
    if (host_cnt[HS_CNT_SURVIVOR_OVERFLOW]) return HS_SPLIT_BATCH;
#ifdef HS_TEST_HOOKS
    if (h->knobs.test_split_above && nq > h->knobs.test_split_above) return HS_SPLIT_BATCH;
#endif
    if (host_cnt[HS_CNT_BAD_QUERY_CODE])
      return fail(h, HS_ERR_INVALID, "residue code outside the alphabet in the queries");
    if (async_items && (host_cnt[8] /* join legality */ || n_items_real > n_items)) {
      ++h->prof.join_async_retries;  // (measurements can exclude such a call: its join ran twice)
      return query_batch(h, d_centers, d_qcodes_ext, nq, q_base, R, brute, d_cand, n_batch_hits, bout, false);
    }
    ms_verify += ev_ms(h, 3, 4);
    if (!brute && n_items) ms_join += ev_ms(h, 11, 10);
    ms_final += ev_ms(h, 4, 5);
    ++launches;
    if (h->knobs.debug_refine)
      fprintf(stderr, "survivors %u -> refined %u -> hits %u\n", host_cnt[0], host_cnt[4], host_cnt[1]);
    if (host_cnt[0] > prov_cap) {
      // the survivor list was too short: once more with room for what the filters reported (64-bit
      // arithmetic: the count may sit just under the overflow flag's 0xF0000000).  The six lists of the
      // exact pass are sized from it -- 48 bytes per entry -- so beyond 2^30 entries, or when the device
      // has no room for them, the batch is cut in halves like a counter overflow instead.
      const uint64_t need = (uint64_t)host_cnt[0] + host_cnt[0] / 8 + 1024;
      if (need > (1ull << 30) && nq > 1) return HS_SPLIT_BATCH;
      if (need > 0xffffffffull) return fail(h, HS_ERR_CAPACITY, "the filter survivors of one query exceed the survivor list");
      const uint64_t have = prov_cap;
      prov_cap = (uint32_t)need;
      if (nq > 1) {  // can the lists grow?  (a failed reserve keeps the old buffer's size at 0: re-reserved below)
        const size_t bytes = (size_t)prov_cap * 8;
        if (h->prov.reserve(bytes) != hipSuccess || h->hit_key.reserve(bytes) != hipSuccess ||
            h->hit_val.reserve(bytes) != hipSuccess) {
          (void)hipGetLastError();
          prov_cap = (uint32_t)have;
          return HS_SPLIT_BATCH;
        }
      }
      continue;
    }
    if (bout) bout->ordered = order_here && !host_cnt[20];
    if (order_here) {
      h->order_failed = host_cnt[20] != 0;
      h->order_failed_R = R;
    }
    break;  // hit_count <= prov_count <= prov_cap <= hit_cap
  }
  h->prof.ms_hash += ev_ms(h, 0, 1);
  if (proj_stats) {
    h->prof.hash_values += (uint64_t)nq * h->LK;
    h->prof.hash_flagged += host_proj[1];
  }
  h->prof.ms_probe += ev_ms(h, 1, 2);
  h->prof.ms_verify += ms_verify;
  h->prof.ms_finalize += ms_final;
  h->prof.verify_launches += launches;
  uint64_t cand_total;
  memcpy(&cand_total, host_cnt + 2, 8);
  h->prof.candidates += brute ? (uint64_t)nq * h->n : cand_total;
  h->prof.provisional += host_cnt[0];
  h->prof.join_batches += n_items ? 1 : 0;
  h->prof.join_i8_batches += (n_items && use_i8) ? 1 : 0;
  if (n_items && use_i8) {
    h->prof.join_row_bytes = (uint32_t)hs_join8_row_bytes(k, wide);
    h->prof.join_wide = (uint32_t)wide;
  }
  h->prof.ms_join += ms_join;
  if (async_items) n_items = n_items_real;
  h->prof.join_items += n_items;
  if (use_i8 && n_items) h->item_cap_hint = n_items + n_items / 4 + 4096;
  if (use_i8 && n_items) ++h->resident_age;
  if (use_i8 && n_items && use_r && h->pin_cnt[42]) {
    h->resident_age = 0;
    h->resident_nq = nq;
    h->resident_share = (double)(h->pin_cnt[42] - h->pin_cnt[41]) / (double)h->pin_cnt[42];
    h->prof.join_items_resident += h->pin_cnt[42] - h->pin_cnt[41];
  }
  if (use_i8 && n_items) {  // sizes the next batch's counter chunks (hs_launch_join8w)
    unsigned long long issued = 0;
    memcpy(&issued, host_cnt + 10, 8);
    h->pairs_per_item = (double)issued / (double)n_items;
  }
  if (use_join) {
    unsigned long long js[2] = {0, 0};
    memcpy(js, host_cnt + 10, 16);  // d_jstats = d_cnt + 10, read back with the counters
    h->prof.join_pairs_issued += js[0];
    h->prof.join_pairs += js[1];
  }
  *n_batch_hits = host_cnt[1];
  return HS_OK;
}

static hs_status run_query(hs_handle* h, const double* d_centers, const uint8_t* d_qcodes, uint64_t nq, double R, bool brute,
                           uint32_t* d_hit_q, uint32_t* d_hit_id, uint32_t* d_hit_table,
                           double* d_hit_dist, uint64_t cap, uint64_t* n_hits, uint64_t* d_cand) {
  if (!h || !n_hits) return HS_ERR_INVALID;
  *n_hits = 0;
  if (!h->built) return fail(h, HS_ERR_STATE, "hs_index_build has not been called");
  if (nq >= (1ull << 27)) return fail(h, HS_ERR_INVALID, "nq must be < 2^27 per call");
  // (a self-join that runs from the residue codes passes no centres)
  if (nq && !d_centers && !d_qcodes && !(h->self_first != HS_NO_SELF && self_codes_ok(h, R))) return HS_ERR_INVALID;
  if (cap && (!d_hit_q || !d_hit_id || !d_hit_dist)) return HS_ERR_INVALID;
  if (!(R == R)) return fail(h, HS_ERR_INVALID, "R is NaN");
  hs_status st = ensure_device(h);
  if (st) return st;
  memset(&h->prof, 0, sizeof(h->prof));
  HS_HIP(h, h->counters.reserve(256));
  HS_HIP(h, hipEventRecord(h->ev[8], h->stream));
  // Centres that are k-mers (every 8 doubles a row of the coordinate table, bit for bit -- what the
  // reference's centres files hold) run from their residue codes, as hs_query_codes's do: the same
  // results from k bytes per query where the point rows are 64 k.  One small kernel and one wait per call.
  if (d_centers && !d_qcodes && nq && h->n && !brute && !h->knobs.no_recognise && h->p.k <= 75) {
    const size_t cb = ((size_t)nq * h->p.k + 15) & ~(size_t)15;
    HS_HIP(h, h->rec_codes.reserve(cb + 16));
    uint32_t* const d_bad = reinterpret_cast<uint32_t*>(h->rec_codes.as<uint8_t>() + cb);
    HS_HIP(h, hipMemsetAsync(d_bad, 0, 4, h->stream));
    HS_HIP(h, hs_launch_recognise_kmers(d_centers, nq, h->p.k, h->coords.as<double>(), h->alphabet,
                                        h->rec_codes.as<uint8_t>(), d_bad, h->stream));
    uint32_t bad = 1;
    HS_HIP(h, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
    if (!bad) {
      d_qcodes = h->rec_codes.as<uint8_t>();
      d_centers = nullptr;
      h->prof.queries_recognised = nq;
    }
  }
  uint64_t total = 0;
  if (nq && h->n && !(brute && R < 0)) {
    // queries per batch: bounds the workspace, which grows with nq * L (2^17 at L >= 8; with few
    // tables -- the one-table indexes of Clustering() -- larger batches, fewer fixed costs)
    uint32_t QB = std::max(1u << 17, std::min(1u << 20, (1u << 20) / h->p.L));
    if (nq > QB && !brute && !h->knobs.query_batch) {
      // More queries than one such batch: as many per batch as a third of the free HBM carries, up to 2^20.
      // The pairs of a batch are (bucket members) x (queries probing the bucket), so the join's operand reuse
      // grows with the batch: at 10^8 k-mers x 32 tables a segment sees ~19 of 125 k queries, ~150 of 10^6.
      // Workspace per query: L x (K + 5 + 18 + 4) words of probe / segment arrays, L x 416 B of gathered
      // query rows + 32 B per work item (~ 1 per probe at worst), its own rows, 16 x 48 B of survivor lists.
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const size_t per_q = (size_t)h->p.L * (4 * ((size_t)h->p.K + 27) + 416 + 64) + 416 + 3 * (size_t)h->d + 768;
        const size_t fit = free_b / 3 / per_q;
        QB = (uint32_t)std::max<size_t>(QB, std::min<size_t>((size_t)1 << 20, fit));
        QB = std::min(QB, (uint32_t)((1ull << 31) / h->p.L) - 1);  // probe numbers carry a flag in bit 31
      }
    }
    if (h->knobs.query_batch) QB = h->knobs.query_batch;  // hs_set_option(HS_OPT_QUERY_BATCH) / HS_QUERY_BATCH
    uint32_t nqb = 0;
    for (uint64_t q0 = 0; q0 < nq; q0 += nqb) {
      nqb = (uint32_t)std::min<uint64_t>(QB, nq - q0);
      uint32_t nh = 0;
      BatchOut bout;
      const uint64_t at = std::min<uint64_t>(total, cap);
      bout.q = d_hit_q + at;
      bout.id = d_hit_id + at;
      bout.table = d_hit_table ? d_hit_table + at : nullptr;
      bout.dist = d_hit_dist + at;
      bout.room = cap - at;
      st = query_batch(h, d_centers ? d_centers + q0 * h->d : nullptr, d_qcodes ? d_qcodes + q0 * h->p.k : nullptr,
                       nqb, (uint32_t)q0, R, brute,
                       d_cand ? d_cand + q0 * h->p.L : nullptr, &nh, cap ? &bout : nullptr);
      if (st == HS_SPLIT_BATCH) {
        // more filter survivors than the 32-bit list counter holds (a radius near the typical
        // distance of bucket mates): the same queries again in batches half the size
        if (nqb == 1) return fail(h, HS_ERR_CAPACITY, "the filter survivors of one query exceed 2^32");
        QB = (nqb + 1) / 2;
        nqb = 0;
        continue;
      }
      if (st) return st;
      if (nh && !bout.ordered) {
        // (brute force, a query with very many hits, HS_SORT_HITS) order of the reference's output
        // by a radix sort on (query, table of first sight, id)
        HS_HIP(h, h->hit_key2.reserve((size_t)nh * 8));
        HS_HIP(h, h->hit_val2.reserve((size_t)nh * 8));
        HS_HIP(h, h->temp.reserve(hs_sort_pairs_u64_u64_temp(nh) + 256));
        HS_HIP(h, hipEventRecord(h->ev[6], h->stream));
        HS_HIP(h, hs_sort_pairs_u64_u64(h->temp.p, h->temp.cap, h->hit_key.as<uint64_t>(),
                                        h->hit_key2.as<uint64_t>(), h->hit_val.as<uint64_t>(),
                                        h->hit_val2.as<uint64_t>(), nh,
                                        37 + bit_width_u32((uint32_t)(q0 + nqb)), h->stream));
        if (total + nh <= cap)
          HS_HIP(h, hs_launch_unpack_hits(h->hit_key2.as<uint64_t>(), h->hit_val2.as<uint64_t>(), nh,
                                          d_hit_q + total, d_hit_id + total,
                                          d_hit_table ? d_hit_table + total : nullptr,
                                          d_hit_dist + total, h->stream));
        HS_HIP(h, hipEventRecord(h->ev[7], h->stream));
        HS_HIP(h, hipStreamSynchronize(h->stream));
        h->prof.ms_finalize += ev_ms(h, 6, 7);
      }
      total += nh;
    }
  } else if (d_cand && nq && !brute) {
    HS_HIP(h, hipMemsetAsync(d_cand, 0, (size_t)nq * h->p.L * 8, h->stream));
  }
  HS_HIP(h, hipEventRecord(h->ev[9], h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  h->prof.ms_total = ev_ms(h, 8, 9);
  h->prof.hits = total;
  *n_hits = total;
  if (total > cap) return fail(h, HS_ERR_CAPACITY, "hit buffers too small; see *n_hits");
  return HS_OK;
}

hs_status hs_query_dev(hs_handle* h, const double* d_centers, uint64_t nq, double R,
                       uint32_t* d_hit_q, uint32_t* d_hit_id, uint32_t* d_hit_table,
                       double* d_hit_dist, uint64_t cap, uint64_t* n_hits, uint64_t* d_cand) {
  return run_query(h, d_centers, nullptr, nq, R, false, d_hit_q, d_hit_id, d_hit_table, d_hit_dist, cap,
                   n_hits, d_cand);
}

hs_status hs_query_codes_dev(hs_handle* h, const uint8_t* d_qcodes, uint64_t nq, double R,
                             uint32_t* d_hit_q, uint32_t* d_hit_id, uint32_t* d_hit_table,
                             double* d_hit_dist, uint64_t cap, uint64_t* n_hits, uint64_t* d_cand) {
  if (nq && !d_qcodes) return HS_ERR_INVALID;
  return run_query(h, nullptr, d_qcodes, nq, R, false, d_hit_q, d_hit_id, d_hit_table, d_hit_dist, cap,
                   n_hits, d_cand);
}

static hs_status host_query(hs_handle* h, const double* centers, const uint8_t* qcodes, uint64_t nq, double R,
                            bool brute, uint32_t* hit_q, uint32_t* hit_id, uint32_t* hit_table, double* hit_dist,
                            uint64_t cap, uint64_t* n_hits, uint64_t* cand) {
  if (!h || !n_hits) return HS_ERR_INVALID;
  if (nq && !centers && !qcodes) return HS_ERR_INVALID;
  hs_status st = ensure_device(h);
  if (st) return st;
  // centres: 8d bytes per query over PCIe; codes: k bytes
  const size_t cbytes = qcodes ? 0 : (size_t)nq * h->d * 8, kbytes = qcodes ? (size_t)nq * h->p.k : 0;
  HS_HIP(h, h->io_centers.reserve(std::max<size_t>(16, cbytes)));
  HS_HIP(h, h->io_codes.reserve(std::max<size_t>(16, kbytes)));
  HS_HIP(h, h->io_q.reserve(std::max<size_t>(16, cap * 4)));
  HS_HIP(h, h->io_id.reserve(std::max<size_t>(16, cap * 4)));
  HS_HIP(h, h->io_table.reserve(std::max<size_t>(16, cap * 4)));
  HS_HIP(h, h->io_dist.reserve(std::max<size_t>(16, cap * 8)));
  if (cand) HS_HIP(h, h->io_cand.reserve(std::max<size_t>(16, (size_t)nq * h->p.L * 8)));
  if (cbytes) HS_HIP(h, hipMemcpyAsync(h->io_centers.p, centers, cbytes, hipMemcpyHostToDevice, h->stream));
  if (kbytes) HS_HIP(h, hipMemcpyAsync(h->io_codes.p, qcodes, kbytes, hipMemcpyHostToDevice, h->stream));
  st = run_query(h, qcodes ? nullptr : h->io_centers.as<double>(), qcodes ? h->io_codes.as<uint8_t>() : nullptr, nq,
                 R, brute, h->io_q.as<uint32_t>(),
                 h->io_id.as<uint32_t>(), h->io_table.as<uint32_t>(), h->io_dist.as<double>(), cap,
                 n_hits, cand ? h->io_cand.as<uint64_t>() : nullptr);
  if (st != HS_OK) return st;
  const uint64_t nh = *n_hits;
  if (nh) {
    HS_HIP(h, hipMemcpyAsync(hit_q, h->io_q.p, nh * 4, hipMemcpyDeviceToHost, h->stream));
    HS_HIP(h, hipMemcpyAsync(hit_id, h->io_id.p, nh * 4, hipMemcpyDeviceToHost, h->stream));
    if (hit_table)
      HS_HIP(h, hipMemcpyAsync(hit_table, h->io_table.p, nh * 4, hipMemcpyDeviceToHost, h->stream));
    HS_HIP(h, hipMemcpyAsync(hit_dist, h->io_dist.p, nh * 8, hipMemcpyDeviceToHost, h->stream));
  }
  if (cand && nq)
    HS_HIP(h, hipMemcpyAsync(cand, h->io_cand.p, (size_t)nq * h->p.L * 8, hipMemcpyDeviceToHost, h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  return HS_OK;
}

hs_status hs_query(hs_handle* h, const double* centers, uint64_t nq, double R, uint32_t* hit_q,
                   uint32_t* hit_id, uint32_t* hit_table, double* hit_dist, uint64_t cap,
                   uint64_t* n_hits, uint64_t* cand) {
  return host_query(h, centers, nullptr, nq, R, false, hit_q, hit_id, hit_table, hit_dist, cap, n_hits, cand);
}

hs_status hs_query_codes(hs_handle* h, const uint8_t* qcodes, uint64_t nq, double R, uint32_t* hit_q,
                         uint32_t* hit_id, uint32_t* hit_table, double* hit_dist, uint64_t cap,
                         uint64_t* n_hits, uint64_t* cand) {
  if (nq && !qcodes) return HS_ERR_INVALID;
  return host_query(h, nullptr, qcodes, nq, R, false, hit_q, hit_id, hit_table, hit_dist, cap, n_hits, cand);
}

// The merge step of the TABLE-partitioned multi-GPU layout (include/hsearch.h): in place on n gathered tuples.
hs_status hs_merge_first_table_dev(hs_handle* h, uint32_t* d_q, uint32_t* d_id, uint32_t* d_table, double* d_dist,
                                   uint64_t n, uint64_t* n_out) {
  if (!h || !n_out) return HS_ERR_INVALID;
  *n_out = 0;
  if (!n) return HS_OK;
  if (!d_q || !d_id || !d_table || !d_dist) return HS_ERR_INVALID;
  if (n >= (1ull << 31)) return fail(h, HS_ERR_INVALID, "more than 2^31 - 1 tuples to merge");
  hs_status st = ensure_device(h);
  if (st) return st;
  const uint32_t n32 = (uint32_t)n;
  HS_HIP(h, h->hit_key.reserve(n * 8));
  HS_HIP(h, h->hit_val.reserve(n * 8));
  HS_HIP(h, h->hit_key2.reserve(n * 8));
  HS_HIP(h, h->hit_val2.reserve(n * 8));
  HS_HIP(h, h->qhits.reserve(2 * (n + 1) * 4));
  HS_HIP(h, h->temp.reserve(std::max(hs_sort_pairs_u64_u64_temp(n), hs_scan_u32_temp(n + 1)) + 256));
  uint64_t *k1 = h->hit_key.as<uint64_t>(), *v1 = h->hit_val.as<uint64_t>();
  uint64_t *k2 = h->hit_key2.as<uint64_t>(), *v2 = h->hit_val2.as<uint64_t>();
  uint32_t* flag = h->qhits.as<uint32_t>();
  uint32_t* pos = flag + (n + 1);
  HS_HIP(h, hs_launch_merge_key1(d_q, d_id, d_table, d_dist, n32, k1, v1, h->stream));
  HS_HIP(h, hs_sort_pairs_u64_u64(h->temp.p, h->temp.cap, k1, k2, v1, v2, n, 64, h->stream));
  HS_HIP(h, hs_launch_merge_flag(k2, n32, flag, h->stream));
  HS_HIP(h, hs_exclusive_scan_u32(h->temp.p, h->temp.cap, flag, pos, n + 1, h->stream));
  HS_HIP(h, hs_launch_merge_compact(k2, v2, pos, n32, k1, v1, h->stream));
  uint32_t kept = 0;
  HS_HIP(h, hipMemcpyAsync(&kept, pos + n, 4, hipMemcpyDeviceToHost, h->stream));
  HS_HIP(h, hipStreamSynchronize(h->stream));
  if (kept) {
    HS_HIP(h, hs_sort_pairs_u64_u64(h->temp.p, h->temp.cap, k1, k2, v1, v2, kept, 64, h->stream));
    HS_HIP(h, hs_launch_unpack_hits(k2, v2, kept, d_q, d_id, d_table, d_dist, h->stream));
    HS_HIP(h, hipStreamSynchronize(h->stream));
  }
  *n_out = kept;
  return HS_OK;
}

hs_status hs_bruteforce(hs_handle* h, const double* centers, uint64_t nq, double R, uint32_t* hit_q,
                        uint32_t* hit_id, double* hit_dist, uint64_t cap, uint64_t* n_hits) {
  return host_query(h, centers, nullptr, nq, R, true, hit_q, hit_id, nullptr, hit_dist, cap, n_hits, nullptr);
}

hs_status hs_self_join(hs_handle* h, double R, int sqrt_test, uint32_t* edge_i, uint32_t* edge_j,
                       uint32_t* edge_table, double* edge_dist, uint64_t cap, uint64_t* n_edges) {
  if (!h) return HS_ERR_INVALID;
  return hs_self_join_range(h, 0, h->n, R, sqrt_test, edge_i, edge_j, edge_table, edge_dist, cap,
                            n_edges);
}

hs_status hs_self_join_range(hs_handle* h, uint64_t first, uint64_t count, double R, int sqrt_test,
                             uint32_t* edge_i, uint32_t* edge_j, uint32_t* edge_table,
                             double* edge_dist, uint64_t cap, uint64_t* n_edges) {
  if (!h || !n_edges) return HS_ERR_INVALID;
  *n_edges = 0;
  if (!h->built) return fail(h, HS_ERR_STATE, "hs_index_build has not been called");
  if (cap && (!edge_i || !edge_j || !edge_dist)) return HS_ERR_INVALID;
  hs_status st = ensure_device(h);
  if (st) return st;
  if (first > h->n || count > h->n - first) return fail(h, HS_ERR_INVALID, "range outside the indexed k-mers");
  const uint64_t n = first + count;
  const uint32_t CH = 1u << 20;  // queries embedded per chunk (8k doubles each): whole batches of run_query
  uint64_t total = 0;
  hs_profile acc;
  memset(&acc, 0, sizeof(acc));
  // the handle's I/O buffers (host-pointer queries use them the same way) and a pinned staging
  // area: Clustering() calls this once per table, reallocating 1.6 GB of centres and faulting in
  // fresh host vectors every time cost more than the join itself
  DevBuf &centers = h->io_centers, &dq = h->io_q, &did = h->io_id, &dt = h->io_table, &dd = h->io_dist;
  const bool sj_timing = h->knobs.cluster_timing;
  auto sj_t0 = std::chrono::steady_clock::now();
  auto sj_lap = [&](const char* what) {
    if (!sj_timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "    sj %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - sj_t0).count());
    sj_t0 = now;
  };
  for (uint64_t q0 = first; q0 < n; q0 += CH) {
    const uint64_t nq = std::min<uint64_t>(CH, n - q0);
    // from the residue codes when every filter on the way can (query_batch's self_codes); embedded
    // centres as for any other query otherwise
    const bool from_codes = self_codes_ok(h, R);
    if (!from_codes) {
      HS_HIP(h, centers.reserve((size_t)nq * h->d * 8));
      sj_lap("centers");
      HS_HIP(h, hs_launch_embed(h->codes.as<uint8_t>() + q0 * h->p.k, nq, (int)h->p.k,
                                h->coords.as<double>(), centers.as<double>(), h->stream));
    }
    uint64_t hcap = std::max<uint64_t>(dq.cap / 4, 3 * nq + 1024), nh = 0;
    for (;;) {
      HS_HIP(h, dq.reserve(hcap * 4));
      HS_HIP(h, did.reserve(hcap * 4));
      HS_HIP(h, dt.reserve(hcap * 4));
      HS_HIP(h, dd.reserve(hcap * 8));
      h->sqrt_test = sqrt_test != 0;
      h->self_first = (uint32_t)q0;  // the pair of a k-mer with itself is dropped on the device
      st = run_query(h, from_codes ? nullptr : centers.as<double>(), nullptr, nq, R, false, dq.as<uint32_t>(), did.as<uint32_t>(),
                     dt.as<uint32_t>(), dd.as<double>(), hcap, &nh, nullptr);
      h->sqrt_test = false;
      h->self_first = HS_NO_SELF;
      if (st == HS_ERR_CAPACITY) {
        hcap = nh + nh / 8 + 1024;
        continue;
      }
      break;
    }
    if (st != HS_OK) return st;
    sj_lap("run_query");
    acc.ms_hash += h->prof.ms_hash; acc.ms_probe += h->prof.ms_probe; acc.ms_verify += h->prof.ms_verify;
    acc.ms_join += h->prof.ms_join; acc.ms_finalize += h->prof.ms_finalize; acc.ms_total += h->prof.ms_total;
    acc.candidates += h->prof.candidates; acc.provisional += h->prof.provisional;
    acc.join_pairs += h->prof.join_pairs; acc.join_pairs_issued += h->prof.join_pairs_issued;
    acc.join_items += h->prof.join_items; acc.join_batches += h->prof.join_batches;
    acc.join_items_resident += h->prof.join_items_resident;
    acc.verify_launches += h->prof.verify_launches;
    acc.hash_values += h->prof.hash_values; acc.hash_flagged += h->prof.hash_flagged;
    HS_HIP(h, h->sj_host.reserve(std::max<size_t>(64, (nh + nh / 8 + 1024) * 20)));
    sj_lap("host buf");
    double* const hd = h->sj_host.as<double>();                       // [nh] doubles first: aligned
    uint32_t* const hq = reinterpret_cast<uint32_t*>(hd + nh);
    uint32_t* const hid = hq + nh;
    uint32_t* const ht = hid + nh;
    if (nh) {
      HS_HIP(h, hipMemcpyAsync(hd, dd.p, nh * 8, hipMemcpyDeviceToHost, h->stream));
      HS_HIP(h, hipMemcpyAsync(hq, dq.p, nh * 4, hipMemcpyDeviceToHost, h->stream));
      HS_HIP(h, hipMemcpyAsync(hid, did.p, nh * 4, hipMemcpyDeviceToHost, h->stream));
      HS_HIP(h, hipMemcpyAsync(ht, dt.p, nh * 4, hipMemcpyDeviceToHost, h->stream));
      HS_HIP(h, hipStreamSynchronize(h->stream));
    }
    sj_lap("d2h");
    for (uint64_t e = 0; e < nh; ++e) {
      const uint64_t i = q0 + hq[e];
      if (i == hid[e]) continue;  // a k-mer is in its own bucket at distance 0
      if (total < cap) {
        edge_i[total] = (uint32_t)i;
        edge_j[total] = hid[e];
        if (edge_table) edge_table[total] = ht[e];
        edge_dist[total] = hd[e];
      }
      ++total;
    }
  }
  h->prof = acc;
  h->prof.hits = total;
  *n_edges = total;
  if (total > cap) return fail(h, HS_ERR_CAPACITY, "edge buffers too small; see *n_edges");
  return HS_OK;
}

hs_status hs_bruteforce_topk(hs_handle* h, const double* centers, uint64_t nq, uint32_t topk,
                             uint32_t* nn_id, double* nn_dist2) {
  if (!h || (nq && (!centers || !nn_id || !nn_dist2)) || topk < 1 || topk > 1024) return HS_ERR_INVALID;
  if (!h->built) return fail(h, HS_ERR_STATE, "hs_index_build has not been called");
  hs_status st = ensure_device(h);
  if (st) return st;
  memset(&h->prof, 0, sizeof(h->prof));
  const int k = (int)h->p.k;
  const uint32_t n = (uint32_t)h->n;
  const uint32_t per_q = (n + HS_SLICE - 1) / HS_SLICE;
  const int n_blocks = h->n_cu * 8;
  HS_HIP(h, h->counters.reserve(256));
  uint32_t* d_cnt = h->counters.as<uint32_t>();
  std::vector<uint64_t> keys, vals;
  std::vector<std::pair<double, uint32_t>> cand;
  const uint32_t QB = 4096;  // queries per batch: two scans of the DB per batch
  for (uint64_t q0 = 0; q0 < nq; q0 += QB) {
    const uint32_t nqb = (uint32_t)std::min<uint64_t>(QB, nq - q0);
    for (uint32_t t = 0; t < nqb * topk; ++t) {
      nn_id[q0 * topk + t] = 0xffffffffu;
      nn_dist2[q0 * topk + t] = INFINITY;
    }
    if (!n) continue;
    HS_HIP(h, h->io_centers.reserve((size_t)nqb * h->d * 8));
    HS_HIP(h, h->tq.reserve((size_t)nqb * k * HS_TROW * 4));
    HS_HIP(h, h->io_misc.reserve((size_t)nqb * per_q * 4 + (size_t)nqb * 4));
    float* d_slice_min = h->io_misc.as<float>();
    float* d_thr = d_slice_min + (size_t)nqb * per_q;
    const double* d_centers = h->io_centers.as<double>();
    HS_HIP(h, hipMemcpyAsync(h->io_centers.p, centers + q0 * h->d, (size_t)nqb * h->d * 8,
                             hipMemcpyHostToDevice, h->stream));
    HS_HIP(h, hs_launch_qtables(d_centers, nqb, k, h->coords.as<double>(), h->alphabet,
                                h->tq.as<float>(), h->stream));
    // pass 1: minimum of every 4096-candidate slice; threshold = k-th smallest slice minimum
    HS_HIP(h, hs_launch_bruteforce(h->packed_all.as<uint4>(), n, h->tq.as<float>(), nqb, k, 0.f, d_cnt,
                                   0, nullptr, nullptr, d_slice_min, n_blocks, h->stream));
    HS_HIP(h, hs_launch_kth_min(d_slice_min, nqb, per_q, topk, d_thr, h->stream));
    // pass 2: everything under the per-query threshold, then exact fp64 distances
    uint32_t prov_cap = (uint32_t)std::max<size_t>(h->prov.cap / 8, std::max<size_t>(1u << 20, 64ull * nqb * topk));
    uint32_t n_prov = 0;
    for (;;) {
      HS_HIP(h, h->prov.reserve((size_t)prov_cap * 8));
      HS_HIP(h, hipMemsetAsync(d_cnt, 0, 8, h->stream));
      HS_HIP(h, hs_launch_bruteforce(h->packed_all.as<uint4>(), n, h->tq.as<float>(), nqb, k, 0.f,
                                     d_cnt, prov_cap, h->prov.as<uint2>(), d_thr, nullptr, n_blocks,
                                     h->stream));
      HS_HIP(h, hipMemcpyAsync(&n_prov, d_cnt, 4, hipMemcpyDeviceToHost, h->stream));
      HS_HIP(h, hipStreamSynchronize(h->stream));
      if (n_prov <= prov_cap) break;
      prov_cap = n_prov + 1024;
    }
    HS_HIP(h, h->hit_key.reserve(std::max<size_t>(16, (size_t)n_prov * 8)));
    HS_HIP(h, h->hit_val.reserve(std::max<size_t>(16, (size_t)n_prov * 8)));
    HS_HIP(h, hs_launch_topk_exact(h->codes.as<uint8_t>(), d_centers, h->coords.as<double>(),
                                   h->prov.as<uint2>(), d_cnt, prov_cap, k, h->hit_key.as<uint64_t>(),
                                   h->hit_val.as<uint64_t>(), h->stream));
    keys.resize(n_prov);
    vals.resize(n_prov);
    if (n_prov) {
      HS_HIP(h, hipMemcpyAsync(keys.data(), h->hit_key.p, (size_t)n_prov * 8, hipMemcpyDeviceToHost, h->stream));
      HS_HIP(h, hipMemcpyAsync(vals.data(), h->hit_val.p, (size_t)n_prov * 8, hipMemcpyDeviceToHost, h->stream));
    }
    HS_HIP(h, hipStreamSynchronize(h->stream));
    // the per-query selection among the few survivors is host work: (d2, id) ascending
    std::vector<std::vector<std::pair<double, uint32_t>>> per(nqb);
    for (uint32_t e = 0; e < n_prov; ++e) {
      double d2;
      memcpy(&d2, &vals[e], 8);
      per[(uint32_t)(keys[e] >> 37)].push_back(std::make_pair(d2, (uint32_t)keys[e]));
    }
    for (uint32_t q = 0; q < nqb; ++q) {
      std::sort(per[q].begin(), per[q].end());
      for (uint32_t t = 0; t < topk && t < per[q].size(); ++t) {
        nn_id[(q0 + q) * topk + t] = per[q][t].second;
        nn_dist2[(q0 + q) * topk + t] = per[q][t].first;
      }
    }
    h->prof.provisional += n_prov;
    h->prof.candidates += 2ull * nqb * n;
  }
  return HS_OK;
}

}  // extern "C"
