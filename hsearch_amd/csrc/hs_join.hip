// hs_join.hip -- bucket-join form of candidate verification (rows a8/a9), gfx950 MFMA.
//
// The reference verifies, for every query q and table l, every member of q's bucket
// (motif_both_points.cpp:232-242).  Many queries probe the SAME bucket (the largest buckets hold
// several per cent of the DB and are probed by thousands of queries), so the work
//     sum over (table, bucket B) of |members(B)| x |queries probing B|
// is a block-sparse product.  Here a workgroup takes a tile of 512 members of one bucket, keeps
// their embedded coordinates as fp16 MFMA A-fragments in registers, and streams the probing
// queries past them in chunks of 32 through LDS: members are read from HBM once per tile instead
// of once per (query, member) pair.
//
// What the MFMA computes is a FILTER, never a result.  Let x1, c1 be member and query restricted
// to the first 4 of the 8 MDS coordinates of every residue position (the table's columns are
// ordered by spread: 4 columns carry 76 % of the mean squared residue distance).  Then
// |x1 - c1|^2 <= |x - c|^2, so with
//     G = |x1|^2 + |c1|^2 - 2 x1^.c1^ - R^2 - slack - e_c |x1|^      (x1^, c1^ = fp16 roundings)
// a pair with exact d2 <= R^2 always has G <= 0:  |2 x1^.c1^ - 2 x1.c1| <= 2 (2^-10 + 2^-22)
// |x1||c1| (fp16 round-to-nearest + Cauchy-Schwarz) is covered by e_c |x1|^ with
// e_c = 1.01 * 2^-9 |c1|, and fp32 accumulation plus the hi/lo splitting of the norms by
// slack (1, plus 0.5 because the kernel tests the sign bit, G < 0).  G is ONE K = 112 GEMM: 100 coordinates + 8 "extras" columns that carry the norms,
// the threshold and the error term (+ 4 pad).  Survivors (G <= 0; a few per million pairs) are
// re-evaluated in the reference's fp64 order and decided by its own test in hs_finalize_kernel,
// exactly as for the streaming kernel.
#include <algorithm>

#include "hs_internal.h"

namespace {

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int JD = 4;       // MDS coordinates per residue used by the filter
constexpr int JG = 14;      // groups of 8 halves: 12 position pairs, position 24 (+pad), extras
constexpr int JK = 8 * JG;  // GEMM depth 112 = 7 k-steps of 16
constexpr int JS = 7;       // k-steps
constexpr int JROW = 120;   // LDS row stride in halves: 240 B = 15 x 16 B -> conflict-free b128
constexpr int JQ = 32;      // queries per chunk (MFMA N)
constexpr int JT = 4;       // 32-member MFMA row tiles per wave
constexpr int JM = 4 * JT * 32;  // members per workgroup tile: 512
constexpr int JQG = 2048;   // queries per work item of the staged fp16 kernel (<= 64 chunks)
constexpr uint32_t JRES = 64;  // survivor slots a wave reserves per atomic
constexpr float JSLACK = 1.5f;  // 1 for the rounding analysis + 0.5 so that "G < 0" covers "G <= 0"

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ---------------------------------------------------------------------------------- query prep
// c16[q] (JK halves): group g < 12 = { -2 c[pos 2g][0..3], -2 c[pos 2g+1][0..3] }, group 12 =
// { -2 c[pos 24][0..3], 0 x4 }, group 13 = extras { 1, 1, hi(v), lo(v), -e_c, 0, 0, 0 } with
// v = |c1|^2 - R^2 - slack and e_c >= 2^-9 * 1.0003 |c1|.  Positions >= k are zero.  One
// wavefront per query.  unsafe |= 1 when fp16 cannot carry the query (the caller then uses the
// streaming kernel for the batch).
__global__ __launch_bounds__(256) void hs_qprep_kernel(const double* __restrict__ centers, uint32_t nq,
                                                       int k, double r2, _Float16* __restrict__ c16,
                                                       uint32_t* __restrict__ unsafe) {
  const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;
  const int lane = lane_id();
  const double* c = centers + (uint64_t)q * 8 * k;
  _Float16* out = c16 + (uint64_t)q * JK;
  double nc = 0.0;
  bool bad = false;
  for (int i = lane; i < JK - 8; i += 64) {  // 104 coordinate slots (the last 4 are pad)
    const int pos = i >> 2, j = i & 3;
    _Float16 h = (_Float16)0.f;
    if (pos < k && pos < 25) {
      const double v = c[8 * pos + j];
      nc += v * v;
      bad = bad || !(fabs(v) < 16000.0);
      h = (_Float16)(float)v;
      h = (_Float16)(-2.0f * (float)h);  // exact: scaling by 2
    }
    out[i] = h;
  }
  // coordinates 4..7 only matter for the legality check
  for (int i = lane; i < 8 * k; i += 64) bad = bad || !(fabs(c[i]) < 16000.0);
  for (int off = 32; off; off >>= 1) nc += __shfl_xor(nc, off);
  bad = bad || !(nc < 30000.0) || !(r2 < 30000.0);
  if (__ballot(bad) && lane == 0) atomicOr(unsafe, 1u);
  if (lane == 0) {
    const float v = (float)(nc - r2 - (double)JSLACK);
    const _Float16 vhi = (_Float16)v;
    const _Float16 vlo = (_Float16)(v - (float)vhi);
    const float ec = (float)(sqrt(nc) * (1.01 / 512.0));
    _Float16* ex = out + 8 * 13;
    ex[0] = (_Float16)1.f;
    ex[1] = (_Float16)1.f;
    ex[2] = vhi;
    ex[3] = vlo;
    ex[4] = (_Float16)(-ec);
    ex[5] = (_Float16)0.f;
    ex[6] = (_Float16)0.f;
    ex[7] = (_Float16)0.f;
  }
}

// fp16 table [32][4] of the first JD coordinates and fp32 squared norms [32] of those coordinates
__global__ void hs_jtables_kernel(const double* __restrict__ coords, int alphabet,
                                  _Float16* __restrict__ tab16, float* __restrict__ rownorm,
                                  uint32_t* __restrict__ unsafe) {
  const int aa = threadIdx.x;
  if (aa >= 32) return;
  double n = 0.0;
  bool bad = false;
  for (int j = 0; j < 8; ++j) {
    const double v = aa < alphabet ? coords[aa * 8 + j] : 0.0;
    bad = bad || !(fabs(v) < 16000.0);
    if (j < JD) {
      n += v * v;
      tab16[aa * JD + j] = (_Float16)(float)v;
    }
  }
  rownorm[aa] = (float)n;
  if (bad) atomicOr(unsafe, 1u);
}

// Probes -> segments by a counting sort on the global bucket number (hs_probe_kernel left, per
// probe, its bucket and its arrival rank inside it):
//   start = exclusive scan of the bucket counts;  sorted_ql[start[bucket] + rank] = probe;
//   the non-empty buckets, in bucket order, are the segments: key = (table << shift) | first sorted
//   position of the bucket, count = probes.  The pseudo-bucket (probes of no bucket) comes last
//   with table = L and is routed nowhere.
// (+ seg_of[p] = segment of sorted position p: what the later per-probe passes index with)
__global__ __launch_bounds__(256) void hs_seg_scatter_kernel(const uint32_t* __restrict__ qbucket,
                                                             const uint32_t* __restrict__ qrank,
                                                             const uint32_t* __restrict__ bucket_start,
                                                             const uint32_t* __restrict__ flag_pos,
                                                             uint32_t nql,
                                                             uint32_t* __restrict__ sorted_ql,
                                                             uint32_t* __restrict__ seg_of) {
  const uint32_t ql = blockIdx.x * 256 + threadIdx.x;
  if (ql >= nql) return;
  const uint32_t gb = qbucket[ql], p = bucket_start[gb] + qrank[ql];
  sorted_ql[p] = ql;
  seg_of[p] = flag_pos[gb];
}
__global__ __launch_bounds__(256) void hs_seg_flag_kernel(const uint32_t* __restrict__ bucket_count,
                                                          uint32_t n, uint32_t* __restrict__ flag) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g <= n) flag[g] = (g < n && bucket_count[g]) ? 1u : 0u;  // flag[n] = 0 closes the scan
}
__global__ __launch_bounds__(256) void hs_seg_emit_kernel(hs_tables_dev tabs,
                                                          const uint32_t* __restrict__ dir_base, int L,
                                                          int shift,
                                                          const uint32_t* __restrict__ bucket_count,
                                                          const uint32_t* __restrict__ flag_pos,
                                                          uint32_t n /* buckets incl. the pseudo one */,
                                                          uint64_t* __restrict__ seg_key,
                                                          uint32_t* __restrict__ seg_cnt,
                                                          uint32_t* __restrict__ n_seg) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g == 0) *n_seg = flag_pos[n];
  if (g >= n) return;
  const uint32_t c = bucket_count[g];
  if (!c) return;
  int l = 0;
  while (l < L && g >= dir_base[l + 1]) ++l;  // dir_base[L] = n - 1 = the pseudo-bucket: l = L
  const uint32_t mstart = l < L ? tabs.t[l].dir_start[g - dir_base[l]] : 0u;
  const uint32_t j = flag_pos[g];
  seg_key[j] = ((uint64_t)(uint32_t)l << shift) | mstart;
  seg_cnt[j] = c;
}

// ---- the same grouping by a sort of the probes (hs_launch_seg_group_sparse)
__global__ __launch_bounds__(256) void hs_iota_kernel(uint32_t n, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = i;
}
// head[p] = 1 where a new bucket starts in the sorted probe list; head[n] = 0 closes the scan
__global__ __launch_bounds__(256) void hs_seg_heads_kernel(const uint32_t* __restrict__ gb, uint32_t n,
                                                           uint32_t* __restrict__ head) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p <= n) head[p] = (p < n && (p == 0 || gb[p] != gb[p - 1])) ? 1u : 0u;
}
// per head: the segment's key (table, first member position) and where its probes start
__global__ __launch_bounds__(256) void hs_seg_emit_sparse_kernel(hs_tables_dev tabs,
                                                                 const uint32_t* __restrict__ dir_base, int L,
                                                                 int shift, const uint32_t* __restrict__ gb,
                                                                 const uint32_t* __restrict__ head,
                                                                 const uint32_t* __restrict__ head_pos,
                                                                 uint32_t n, uint64_t* __restrict__ seg_key,
                                                                 uint32_t* __restrict__ seg_start,
                                                                 uint32_t* __restrict__ n_seg,
                                                                 uint32_t* __restrict__ seg_of) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p == 0) {
    *n_seg = head_pos[n];
    seg_start[head_pos[n]] = n;
  }
  if (p < n) seg_of[p] = head_pos[p] + head[p] - 1u;  // heads before p, p's own included: its segment + 1
  if (p >= n || !head[p]) return;
  const uint32_t g = gb[p];
  int l = 0;
  while (l < L && g >= dir_base[l + 1]) ++l;  // dir_base[L] = the pseudo-bucket: l = L
  const uint32_t mstart = l < L ? tabs.t[l].dir_start[g - dir_base[l]] : 0u;
  const uint32_t j = head_pos[p];
  seg_key[j] = ((uint64_t)(uint32_t)l << shift) | mstart;
  seg_start[j] = p;
}
__global__ __launch_bounds__(256) void hs_seg_counts_kernel(const uint32_t* __restrict__ seg_start,
                                                            const uint32_t* __restrict__ n_seg,
                                                            uint32_t* __restrict__ seg_cnt) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j < *n_seg) seg_cnt[j] = seg_start[j + 1] - seg_start[j];
}

// Routing: a segment (bucket x its probing queries) goes to the MFMA join when enough queries share
// it, otherwise its queries stay with the streaming kernel (one wavefront per query and slice).
// items[j] = member tiles x query groups for joined segments, 0 otherwise / past the end.
__global__ __launch_bounds__(256) void hs_seg_route_kernel(const uint64_t* __restrict__ seg_key,
                                                           const uint32_t* __restrict__ seg_cnt,
                                                           const uint32_t* __restrict__ seg_qoff,
                                                           const uint32_t* __restrict__ n_seg,
                                                           const uint32_t* __restrict__ sorted_ql,
                                                           const uint32_t* __restrict__ qcount,
                                                           uint32_t n_max, uint32_t min_q,
                                                           uint32_t min_m, uint32_t jm, int L, int shift,
                                                           uint32_t max_q_resident,
                                                           uint32_t* __restrict__ items,
                                                           unsigned long long* __restrict__ stats) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  uint32_t it = 0;
  unsigned long long issued = 0, real = 0;
  const uint32_t jqg = jm < HS_JM_BLOCK ? HS_JQG_WAVE : (uint32_t)JQG;
  if (j <= n_max && j < *n_seg && (seg_key[j] >> shift) < (uint64_t)L) {
    const uint32_t m = qcount[sorted_ql[seg_qoff[j]]];
    const uint32_t nq = seg_cnt[j];
    // thin segments (few probing queries or few members) go to the per-pair filter; a BIG bucket goes
    // to the join whatever its query count: its mostly empty query tile costs next to nothing there,
    // while the per-pair filter would walk its thousands of members in one chain of dependent loads
    if ((nq >= min_q && m >= min_m) || m >= 512u) {
      it = ((m + jm - 1) / jm) * ((nq + jqg - 1) / jqg);
      // MFMA pairs actually issued (128-row waves x 32-column chunks) vs real pairs
      const uint32_t rm = jm < HS_JM_BLOCK ? jm : 32u * JT;  // rows of one wave's member tile
      // (the query-resident kernel issues 16-query column tiles, the others 32-query tiles)
      const uint32_t cq = nq <= max_q_resident ? 16u : (uint32_t)JQ;
      issued = (unsigned long long)((m + rm - 1) / rm * rm) * ((nq + cq - 1) / cq * cq);
      real = (unsigned long long)m * nq;
    }
  }
  if (j <= n_max) items[j] = it;
  // the two statistics: one pair of global atomics per block (same-address atomics are slow)
  for (int off = 32; off; off >>= 1) {
    issued += __shfl_xor(issued, off);
    real += __shfl_xor(real, off);
  }
  __shared__ unsigned long long s_st[4][2];
  if ((threadIdx.x & 63) == 0) {
    s_st[threadIdx.x >> 6][0] = issued;
    s_st[threadIdx.x >> 6][1] = real;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long a = s_st[0][0] + s_st[1][0] + s_st[2][0] + s_st[3][0];
    const unsigned long long b = s_st[0][1] + s_st[1][1] + s_st[2][1] + s_st[3][1];
    if (a) atomicAdd(stats + 0, a);
    if (b) atomicAdd(stats + 1, b);
  }
}

// nslices[ql] = 0 for probes whose segment was routed to the join (they keep their slice count,
// written by the probe kernel, otherwise).  One thread per sorted probe; segment by binary search.
__global__ __launch_bounds__(256) void hs_seg_unslice_kernel(const uint32_t* __restrict__ seg_of,
                                                             const uint32_t* __restrict__ items,
                                                             const uint32_t* __restrict__ sorted_ql,
                                                             uint32_t nql, uint32_t* __restrict__ nslices) {
  const uint32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= nql) return;
  if (items[seg_of[p]]) nslices[sorted_ql[p]] = 0;
}

// Item numbering order of the segments: those with many probing queries first (a stable
// two-class partition).  The join hands items out from a counter, so the last ones handed out
// should be small.  big[j] -> exclusive scan -> order[] and the item counts in that order.
// Item numbering order of the segments, three classes, stable inside each: many-query segments first
// (>= min_q probing queries: the long items that should start early), then the rest, and LAST the
// segments with at most max_q_resident probing queries (0: no such class) -- their items form the tail
// [split, total) of the item list, which the query-resident join kernel (hs_join8r_kernel) takes while
// the head goes to the query-streaming one.
__global__ __launch_bounds__(256) void hs_seg_big_kernel(const uint32_t* __restrict__ seg_cnt,
                                                         const uint32_t* __restrict__ items,
                                                         uint32_t n, uint32_t min_q, uint32_t max_q_resident,
                                                         uint32_t* __restrict__ big,
                                                         uint32_t* __restrict__ res) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t c = seg_cnt[j];
  const bool has = items[j] != 0u;
  big[j] = (has && c >= min_q) ? 1u : 0u;
  if (res) res[j] = (has && c <= max_q_resident && c < min_q) ? 1u : 0u;
}
// big_pos / res_pos have n + 1 entries: [n] = number of segments of the class (res_pos may be null)
__global__ __launch_bounds__(256) void hs_seg_order_kernel(const uint32_t* __restrict__ big_pos,
                                                           const uint32_t* __restrict__ res_pos,
                                                           const uint32_t* __restrict__ items,
                                                           uint32_t n, uint32_t* __restrict__ order,
                                                           uint32_t* __restrict__ items_ordered) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const bool big = big_pos[j + 1] != big_pos[j];
  const bool res = res_pos && res_pos[j + 1] != res_pos[j];
  const uint32_t n_big = big_pos[n], n_res = res_pos ? res_pos[n] : 0u;
  const uint32_t before_res = res_pos ? res_pos[j] : 0u;
  const uint32_t pos = big ? big_pos[j]
                       : res ? (n - n_res) + before_res
                             : n_big + (j - big_pos[j] - before_res);
  order[pos] = j;
  items_ordered[pos] = items[j];
}
// split[0] = first item of the query-resident class, split[1] = number of items
__global__ void hs_item_split_kernel(const uint32_t* __restrict__ item_off, const uint32_t* __restrict__ res_pos,
                                     uint32_t n, uint32_t* __restrict__ split) {
  if (threadIdx.x || blockIdx.x) return;
  const uint32_t n_res = res_pos ? res_pos[n] : 0u;
  split[0] = item_off[n - n_res];
  split[1] = item_off[n];
}

// One descriptor (2 x uint4) per work item:
//   { offset of the bucket's first packed member from table 0's packed array (lo, hi), M, tile },
//   { qoff, q_begin, q_end, first sorted position of the bucket }
// One thread per item.  The item's segment = the largest position j of `order` with item_off[j] <= item
// (zero-item positions share their successor's offset, so the last such j owns the item): a block's 256
// consecutive items mostly span a few hundred positions, so thread 0 finds the block's first position by a
// binary search over item_off in memory, the block copies the next positions' offsets to LDS, and every
// thread searches there; a block whose items span more positions than the window holds (a run of zero-item
// positions inside it) falls back to the search in memory.  (One thread per SEGMENT, writing its items in
// a loop, was measured slower: 32-byte stores scattered over the descriptor array.)
__global__ __launch_bounds__(256) void hs_item_desc_kernel(hs_tables_dev tabs,
                                                           const uint64_t* __restrict__ seg_key,
                                                           const uint32_t* __restrict__ seg_cnt,
                                                           const uint32_t* __restrict__ seg_qoff,
                                                           const uint32_t* __restrict__ item_off,
                                                           uint32_t n_max,
                                                           const uint32_t* __restrict__ sorted_ql,
                                                           const uint32_t* __restrict__ qcount,
                                                           uint32_t n_items, uint32_t jm, int shift,
                                                           const uint32_t* __restrict__ order, int PW,
                                                           const uint32_t* __restrict__ n_items_dev,
                                                           uint4* __restrict__ desc) {
  constexpr uint32_t WIN = 768;
  __shared__ uint32_t s_off[WIN + 1];
  __shared__ uint32_t s_first;
  const uint32_t item0 = blockIdx.x * 256, item = item0 + threadIdx.x;
  if (n_items_dev) n_items = min(n_items, *n_items_dev);  // n_items = capacity of desc then
  if (item0 >= n_items) return;
  if (threadIdx.x == 0) {
    uint32_t lo = 0, hi = n_max;  // largest j with item_off[j] <= item0
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (item_off[mid] <= item0) lo = mid; else hi = mid;
    }
    s_first = lo;
  }
  __syncthreads();
  const uint32_t first = s_first;
  for (uint32_t i = threadIdx.x; i <= WIN; i += 256) s_off[i] = first + i <= n_max ? item_off[first + i] : 0xffffffffu;
  __syncthreads();
  if (item >= n_items) return;
  uint32_t lo;
  if (s_off[WIN] > item) {  // the item's position lies inside the window
    uint32_t l = 0, h = WIN;
    while (h - l > 1) {
      const uint32_t mid = (l + h) >> 1;
      if (s_off[mid] <= item) l = mid; else h = mid;
    }
    lo = first + l;
  } else {
    lo = first;
    uint32_t hi = n_max;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (item_off[mid] <= item) lo = mid; else hi = mid;
    }
  }
  const uint32_t seg = order[lo];  // item_off runs over the segments in `order`
  const uint64_t key = seg_key[seg];
  const uint32_t nQ = seg_cnt[seg], qoff = seg_qoff[seg];
  const uint32_t M = qcount[sorted_ql[qoff]];
  const uint32_t jqg = jm < HS_JM_BLOCK ? HS_JQG_WAVE : (uint32_t)JQG;
  const uint32_t tiles_m = (M + jm - 1) / jm;
  const uint32_t local = item - item_off[lo];
  const uint32_t mt = local % tiles_m, qg = local / tiles_m;
  const uint32_t q_begin = qg * jqg;
  const uint32_t mstart = (uint32_t)(key & ((1ull << shift) - 1ull));
  // ENTRY number of the segment's first member counted from table 0's first entry (an entry = PW
  // 16-byte words of the packed array, one record of the record array): base + offset keeps the
  // member loads in the global address space (a pointer rebuilt from integers compiles to flat
  // loads, whose lgkmcnt accounting stalls the LDS waits of the MFMA loop)
  const int64_t off = (reinterpret_cast<intptr_t>(tabs.t[(uint32_t)(key >> shift)].packed) -
                       reinterpret_cast<intptr_t>(tabs.t[0].packed)) / (16 * (int64_t)PW) + (int64_t)mstart;
  desc[2 * (uint64_t)item] = make_uint4((uint32_t)(uint64_t)off, (uint32_t)((uint64_t)off >> 32), M, mt);
  desc[2 * (uint64_t)item + 1] = make_uint4(qoff, q_begin, min(nQ, q_begin + jqg), mstart);
}

// c16 rows in SEGMENT order, so that a chunk of 32 probing queries is one contiguous 7 KB block
__global__ __launch_bounds__(256) void hs_gather_c16_kernel(const _Float16* __restrict__ c16,
                                                            const uint32_t* __restrict__ sorted_ql,
                                                            uint32_t nql, int L,
                                                            _Float16* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (uint64_t)nql * JG) return;
  const uint32_t p = (uint32_t)(t / JG);
  const int g = (int)(t - (uint64_t)p * JG);
  const uint32_t q = sorted_ql[p] / (uint32_t)L;
  *reinterpret_cast<uint4*>(out + (uint64_t)p * JK + g * 8) =
      *reinterpret_cast<const uint4*>(c16 + (uint64_t)q * JK + g * 8);
}

// ------------------------------------------------------------------------------------------ join
// residue at bit BIT of a 128-bit packed word (x = bits 0..31, ...), BIT + 5 <= 128
template <int BIT>
__device__ __forceinline__ uint32_t residue_at(uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
  constexpr int wi = BIT >> 5, sh = BIT & 31;
  const uint32_t lo = wi == 0 ? x : wi == 1 ? y : wi == 2 ? z : w;
  if constexpr (sh > 27) {
    const uint32_t hi = wi == 0 ? y : wi == 1 ? z : w;
    return __funnelshift_r(lo, hi, sh) & 31u;
  } else {
    return (lo >> sh) & 31u;
  }
}

constexpr int JPIECES = JQ * JG;  // 16-byte pieces of one query chunk: 448

__device__ __forceinline__ half8 lds_b(const _Float16* tile, int off) {
  return *reinterpret_cast<const half8*>(tile + off);
}

// A fragments (whole K) of one 32-member row tile for lane (r, h): k-step s < 6 carries positions
// 4s + 2h and 4s + 2h + 1; k-step 6 carries position 24 (lower half) / the extras (upper half).
__device__ __forceinline__ void build_afrags(const uint4 pk, int h, int k, const _Float16* sTab,
                                             const float* sNorm, half8 (&A)[JS]) {
  // lanes of the upper half take positions 2,3, 6,7, ...: shift the 125-bit word down by 10
  const uint32_t sh = 10u * (uint32_t)h;
  const uint32_t x = __funnelshift_r(pk.x, pk.y, sh), y = __funnelshift_r(pk.y, pk.z, sh),
                 z = __funnelshift_r(pk.z, pk.w, sh), w = pk.w >> sh;
  float nx = 0.f;
#define HS_AFRAG(S)                                                                 \
  {                                                                                 \
    const uint32_t a0 = residue_at<20 * S>(x, y, z, w), a1 = residue_at<20 * S + 5>(x, y, z, w); \
    const half4 lo = *reinterpret_cast<const half4*>(&sTab[a0 * JD]);               \
    const half4 hi = *reinterpret_cast<const half4*>(&sTab[a1 * JD]);               \
    A[S] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);                 \
    nx += (4 * S + 2 * h < k) ? sNorm[a0] : 0.f;                                    \
    nx += (4 * S + 2 * h + 1 < k) ? sNorm[a1] : 0.f;                                \
  }
  HS_AFRAG(0) HS_AFRAG(1) HS_AFRAG(2) HS_AFRAG(3) HS_AFRAG(4) HS_AFRAG(5)
#undef HS_AFRAG
  // position 24 sits at bit 120 of the UNSHIFTED word
  const uint32_t a24 = (pk.w >> 24) & 31u;
  if (h == 0) nx += (24 < k) ? sNorm[a24] : 0.f;
  nx += __shfl_xor(nx, 32);  // |x1|^2 = this half's positions + the other half's
  const half4 p24 = *reinterpret_cast<const half4*>(&sTab[a24 * JD]);
  const half4 zero4 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
  const _Float16 nhi = (_Float16)nx;
  const _Float16 nlo = (_Float16)(nx - (float)nhi);
  const _Float16 xn = (_Float16)(sqrtf(nx) * 1.002f);
  half8 ex;
  ex[0] = nhi; ex[1] = nlo; ex[2] = (_Float16)1.f; ex[3] = (_Float16)1.f;
  ex[4] = xn; ex[5] = (_Float16)0.f; ex[6] = (_Float16)0.f; ex[7] = (_Float16)0.f;
  A[6] = h ? ex : __builtin_shufflevector(p24, zero4, 0, 1, 2, 3, 4, 5, 6, 7);
}

// One work item = (member tile of <= 512 bucket entries, group of <= 2048 probing queries).
// Waves: 4 x (4 x 32 members); A fragments (whole K) live in registers for the item; query chunks
// of 32 stream through a double-buffered LDS tile, prefetched into registers one chunk ahead; the
// descriptor, packed members and first query chunk of the NEXT item are fetched while the current
// item computes, so an item's prologue is only the A-fragment build (LDS table reads).
__global__ __launch_bounds__(256, 2) void hs_join_kernel(
    const uint4* __restrict__ desc, uint32_t n_items, const uint4* __restrict__ packed_base,
    const uint32_t* __restrict__ sorted_ql, const _Float16* __restrict__ c16s,
    const _Float16* __restrict__ tab16, const float* __restrict__ rownorm, int k,
    uint32_t* __restrict__ prov_count, uint32_t prov_cap, uint2* __restrict__ prov) {
  __shared__ __attribute__((aligned(16))) _Float16 sB[2][JQ * JROW];
  __shared__ __attribute__((aligned(16))) _Float16 sTab[32 * JD];
  __shared__ float sNorm[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  if (tid < 32 * JD) sTab[tid] = tab16[tid];
  if (tid < 32) sNorm[tid] = rownorm[tid];
  __syncthreads();
  // This thread's 2 pieces of a chunk (the last 64 threads repeat piece 447: same bytes to the
  // same LDS address, which keeps every load and store unconditional).
  int src_piece[2], dst[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = min(tid + 256 * j, JPIECES - 1);
    const int row = i / JG, g = i - row * JG;
    src_piece[j] = i;
    dst[j] = row * JROW + g * 8;
  }
  const int boff = r * JROW + h * 8;  // B fragment of k-step s: boff + 16 s (halves)
  int buf = 0;
  uint32_t item = blockIdx.x;
  if (item >= n_items) return;
  uint32_t res_base = 0, res_used = JRES;  // no survivor block reserved yet
  // the two workgroups of a CU run the same loop: start the second half a chunk later so that one
  // group's MFMA phase tends to face the other's load/LDS/barrier phase
  if (blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_sleep(8);
  uint4 d0 = desc[2 * (uint64_t)item], d1 = desc[2 * (uint64_t)item + 1];
  uint4 pk[JT], pre0, pre1;
  {
    const uint4* packed = packed_base + (int64_t)(((uint64_t)d0.y << 32) | (uint64_t)d0.x);
    const uint32_t idx = d0.w * JM + wave * (32 * JT) + r;
#pragma unroll
    for (int t = 0; t < JT; ++t) pk[t] = packed[min(idx + 32 * t, d0.z - 1)];
    const uint4* src = reinterpret_cast<const uint4*>(c16s + (uint64_t)(d1.x + d1.y) * JK);
    pre0 = src[src_piece[0]];
    pre1 = src[src_piece[1]];
  }
  while (true) {
    const uint32_t M = d0.z, mt = d0.w;
    const uint32_t qoff = d1.x, q_begin = d1.y, q_end = d1.z, mstart = d1.w;
    const uint32_t wbase = mt * JM + wave * (32 * JT);
    const bool wave_on = wbase < M;  // wave-uniform: this wave owns at least one real member
    const uint32_t next_item = item + gridDim.x;
    const bool has_next = next_item < n_items;
    uint4 nd0 = d0, nd1 = d1;
    if (has_next) {
      nd0 = desc[2 * (uint64_t)next_item];
      nd1 = desc[2 * (uint64_t)next_item + 1];
    }
    half8 A[JT][JS];
    if (wave_on) {
#pragma unroll
      for (int t = 0; t < JT; ++t) build_afrags(pk[t], h, k, sTab, sNorm, A[t]);
    }
    for (uint32_t qc = q_begin; qc < q_end; qc += JQ) {
      _Float16* tile = sB[buf];
      *reinterpret_cast<uint4*>(&tile[dst[0]]) = pre0;
      *reinterpret_cast<uint4*>(&tile[dst[1]]) = pre1;
      __syncthreads();  // tile complete; also: every wave is past its reads of the other buffer
      {
        // next chunk of this item, or the first chunk of the next item (all uniform)
        const bool more = qc + JQ < q_end;
        const uint64_t row = more ? (uint64_t)(qoff + qc + JQ) : (uint64_t)(nd1.x + nd1.y);
        const uint4* src = reinterpret_cast<const uint4*>(c16s + row * JK);
        pre0 = src[src_piece[0]];
        pre1 = src[src_piece[1]];
      }
      if (qc == q_begin) {  // the A fragments are built: the packed words can be replaced
        const uint4* packed = packed_base + (int64_t)(((uint64_t)nd0.y << 32) | (uint64_t)nd0.x);
        const uint32_t idx = nd0.w * JM + wave * (32 * JT) + r;
#pragma unroll
        for (int t = 0; t < JT; ++t) pk[t] = packed[min(idx + 32 * t, nd0.z - 1)];
      }
      buf ^= 1;
      if (!wave_on) continue;
      floatx16 acc[JT];
#pragma unroll
      for (int t = 0; t < JT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
      // B fragments ping-pong between two register sets; the read for k-step s+2 is issued right
      // behind the 4 MFMAs of k-step s, so its LDS latency runs under the MFMAs of k-step s+1.
      half8 b0 = lds_b(tile, boff), b1 = lds_b(tile, boff + 16);
#define HS_STEP(S, B)                                                                        \
  _Pragma("unroll") for (int t = 0; t < JT; ++t)                                             \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[t][S], B, acc[t], 0, 0, 0);          \
  if (S + 2 < JS) B = lds_b(tile, boff + 16 * (S + 2));
      HS_STEP(0, b0) HS_STEP(1, b1) HS_STEP(2, b0) HS_STEP(3, b1) HS_STEP(4, b0) HS_STEP(5, b1)
      HS_STEP(6, b0)
#undef HS_STEP
      // pin that order: 2 reads, 5 x (4 MFMA, 1 read), 8 MFMA  (0x100 = DS read, 0x008 = MFMA)
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int s = 0; s < JS - 2; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, JT, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * JT, 0);
      // ---- survivors: G <= 0 (a few per million).  D layout: col = lane & 31,
      //      row = (i & 3) + 8 (i >> 2) + 4 h.
      // "some G < 0" = sign bit of the OR of all 64 accumulators: a log-depth tree of v_or3
      // instead of a 64-deep chain of dependent v_min (which cost 17 % of the kernel).  The test
      // is G < 0 (sign bit) rather than G <= 0; JSLACK carries the half unit that makes up for it.
      uint32_t sg[JT];
#pragma unroll
      for (int t = 0; t < JT; ++t) {
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o[j] = __float_as_uint(acc[t][4 * j]) | __float_as_uint(acc[t][4 * j + 1]) |
                 __float_as_uint(acc[t][4 * j + 2]) | __float_as_uint(acc[t][4 * j + 3]);
        sg[t] = (o[0] | o[1]) | (o[2] | o[3]);
      }
      uint32_t sany = sg[0];
#pragma unroll
      for (int t = 1; t < JT; ++t) sany |= sg[t];
      if (__ballot((int)sany < 0)) {
        const bool col_ok = qc + (uint32_t)r < q_end;
        const uint32_t ql = HS_PROV_INDIRECT | (qoff + qc + (uint32_t)r);
#pragma unroll
        for (int t = 0; t < JT; ++t) {
          uint32_t mask = 0;
#pragma unroll
          for (int i = 0; i < 16; ++i) mask |= (__float_as_uint(acc[t][i]) >> 31) << i;
          if (!col_ok) mask = 0;
          while (__ballot(mask != 0)) {
            uint32_t idx = 0;
            bool pass = false;
            if (mask) {
              const int i = __ffs((int)mask) - 1;
              mask &= mask - 1;
              idx = wbase + (uint32_t)(t * 32 + (i & 3) + 8 * (i >> 2) + 4 * h);
              pass = idx < M;
            }
            const unsigned long long m = __ballot(pass);
            if (m) {
              // survivor slots are reserved 64 at a time per wave: one atomic per ~64 survivors
              // instead of one per ballot (a single counter saturates near 88 atomics/us)
              const uint32_t cnt = (uint32_t)__popcll(m);
              if (res_used + cnt > JRES) {
                if (res_used < JRES && lane >= (int)res_used && res_base + lane < prov_cap)
                  prov[res_base + lane] = make_uint2(0xffffffffu, 0u);  // unused tail: skipped later
                uint32_t base = 0;
                if (lane == 0) base = hs_reserve_survivors(prov_count, (uint32_t)JRES);
                res_base = __builtin_amdgcn_readfirstlane(base);
                res_used = 0;
              }
              if (pass) {
                const uint32_t o = res_base + res_used + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (o < prov_cap) prov[o] = make_uint2(ql, mstart + idx);
              }
              res_used += cnt;
            }
          }
        }
      }
    }
    if (!has_next) break;
    item = next_item;
    d0 = nd0;
    d1 = nd1;
  }
  if (res_used < JRES && lane >= (int)res_used && res_base + lane < prov_cap)
    prov[res_base + lane] = make_uint2(0xffffffffu, 0u);
}

inline unsigned blocks_for(uint64_t n, unsigned per = 256) { return (unsigned)((n + per - 1) / per); }

}  // namespace

hipError_t hs_launch_jtables(const double* d_coords, int alphabet, void* d_tab16, float* d_rownorm,
                             uint32_t* d_unsafe, hipStream_t s) {
  hs_jtables_kernel<<<1, 32, 0, s>>>(d_coords, alphabet, (_Float16*)d_tab16, d_rownorm, d_unsafe);
  return hipGetLastError();
}

hipError_t hs_launch_qprep(const double* d_centers, uint32_t nq, int k, double r2, void* d_c16,
                           uint32_t* d_unsafe, hipStream_t s) {
  if (!nq) return hipSuccess;
  hs_qprep_kernel<<<blocks_for(nq, 4), 256, 0, s>>>(d_centers, nq, k, r2, (_Float16*)d_c16, d_unsafe);
  return hipGetLastError();
}

hipError_t hs_launch_seg_group(const hs_tables_dev& tabs, const uint32_t* d_dir_base, int L, int shift,
                               uint32_t nb_total, const uint32_t* d_bucket_count,
                               uint32_t* d_bucket_work, void* d_temp, size_t temp_bytes,
                               const uint32_t* d_qbucket, const uint32_t* d_qrank, uint32_t nql,
                               uint32_t* d_sorted_ql, uint64_t* d_seg_key, uint32_t* d_seg_cnt,
                               uint32_t* d_n_seg, uint32_t* d_seg_of, hipStream_t s) {
  if (!nql) return hipSuccess;
  const uint32_t n = nb_total + 1;  // buckets incl. the pseudo one; arrays have n + 1 entries
  uint32_t* start = d_bucket_work;
  uint32_t* flag = d_bucket_work + (n + 1);
  uint32_t* flag_pos = d_bucket_work + 2 * (size_t)(n + 1);
  hipError_t e = hs_exclusive_scan_u32(d_temp, temp_bytes, d_bucket_count, start, (size_t)n + 1, s);
  if (e != hipSuccess) return e;
  hs_seg_flag_kernel<<<blocks_for((uint64_t)n + 1), 256, 0, s>>>(d_bucket_count, n, flag);
  e = hs_exclusive_scan_u32(d_temp, temp_bytes, flag, flag_pos, (size_t)n + 1, s);
  if (e != hipSuccess) return e;
  hs_seg_scatter_kernel<<<blocks_for(nql), 256, 0, s>>>(d_qbucket, d_qrank, start, flag_pos, nql, d_sorted_ql, d_seg_of);
  hs_seg_emit_kernel<<<blocks_for(n), 256, 0, s>>>(tabs, d_dir_base, L, shift, d_bucket_count, flag_pos,
                                                   n, d_seg_key, d_seg_cnt, d_n_seg);
  return hipGetLastError();
}

// Bucket partition: most probes of a batch find nothing (their bucket belongs to another part).  The ones
// that did, in probe order: flag -> exclusive scan -> (bucket, probe) pairs; the grouping then sorts and walks
// those alone (d_pos[nql] = their number).
// (list != null: the n candidates are the probes list[0 .. n), ascending; else the probes 0 .. n - 1)
__global__ __launch_bounds__(256) void hs_found_flags_kernel(const uint32_t* __restrict__ qbucket,
                                                             const uint32_t* __restrict__ list, uint32_t n,
                                                             uint32_t nb_total, uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i <= n) flag[i] = i < n && qbucket[list ? list[i] : i] != nb_total;
}
__global__ __launch_bounds__(256) void hs_found_scatter_kernel(const uint32_t* __restrict__ qbucket,
                                                               const uint32_t* __restrict__ list, uint32_t n,
                                                               uint32_t nb_total, const uint32_t* __restrict__ pos,
                                                               uint32_t* __restrict__ keys, uint32_t* __restrict__ probes) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t ql = list ? list[i] : i;
  const uint32_t gb = qbucket[ql];
  if (gb == nb_total) return;
  keys[pos[i]] = gb;
  probes[pos[i]] = ql;
}
hipError_t hs_launch_found_probes(const uint32_t* d_qbucket, uint32_t nql, uint32_t nb_total, void* d_temp,
                                  size_t temp_bytes, uint32_t* d_flag, uint32_t* d_pos, uint32_t* d_keys,
                                  uint32_t* d_probes, hipStream_t s, const uint32_t* d_list) {
  if (!nql) return hipSuccess;
  hs_found_flags_kernel<<<blocks_for((uint64_t)nql + 1), 256, 0, s>>>(d_qbucket, d_list, nql, nb_total, d_flag);
  hipError_t e = hs_exclusive_scan_u32(d_temp, temp_bytes, d_flag, d_pos, (size_t)nql + 1, s);
  if (e != hipSuccess) return e;
  hs_found_scatter_kernel<<<blocks_for(nql), 256, 0, s>>>(d_qbucket, d_list, nql, nb_total, d_pos, d_keys, d_probes);
  return hipGetLastError();
}

// d_probes_in == null: the probes are 0 .. nql - 1 (d_iota is filled with them); else the nql probe numbers
// that belong to the keys d_qbucket[0 .. nql) (hs_launch_found_probes)
hipError_t hs_launch_seg_group_sparse(const hs_tables_dev& tabs, const uint32_t* d_dir_base, int L, int shift,
                                      uint32_t nb_total, void* d_temp, size_t temp_bytes,
                                      const uint32_t* d_qbucket, uint32_t* d_keys_sorted, uint32_t* d_iota,
                                      uint32_t* d_work, uint32_t nql, uint32_t* d_sorted_ql,
                                      uint64_t* d_seg_key, uint32_t* d_seg_cnt, uint32_t* d_n_seg,
                                      uint32_t* d_seg_of, hipStream_t s, const uint32_t* d_probes_in) {
  if (!nql) return hipSuccess;
  uint32_t* head = d_work;
  uint32_t* head_pos = d_work + ((size_t)nql + 1);
  uint32_t* seg_start = d_work + 2 * ((size_t)nql + 1);
  int bits = 1;
  while (bits < 32 && (nb_total >> bits)) ++bits;  // bucket numbers are <= nb_total
  if (!d_probes_in) hs_iota_kernel<<<blocks_for(nql), 256, 0, s>>>(nql, d_iota);
  hipError_t e = hs_sort_pairs_u32_u32(d_temp, temp_bytes, d_qbucket, d_keys_sorted, d_probes_in ? d_probes_in : d_iota,
                                       d_sorted_ql, nql, bits, s);
  if (e != hipSuccess) return e;
  hs_seg_heads_kernel<<<blocks_for((uint64_t)nql + 1), 256, 0, s>>>(d_keys_sorted, nql, head);
  e = hs_exclusive_scan_u32(d_temp, temp_bytes, head, head_pos, (size_t)nql + 1, s);
  if (e != hipSuccess) return e;
  hs_seg_emit_sparse_kernel<<<blocks_for(nql), 256, 0, s>>>(tabs, d_dir_base, L, shift, d_keys_sorted, head,
                                                            head_pos, nql, d_seg_key, seg_start, d_n_seg, d_seg_of);
  hs_seg_counts_kernel<<<blocks_for(nql), 256, 0, s>>>(seg_start, d_n_seg, d_seg_cnt);
  return hipGetLastError();
}

hipError_t hs_launch_seg_route(const uint64_t* d_seg_key, const uint32_t* d_seg_cnt,
                               const uint32_t* d_seg_qoff, const uint32_t* d_n_seg,
                               const uint32_t* d_sorted_ql, const uint32_t* d_qcount, uint32_t n_max,
                               uint32_t min_q, uint32_t min_m, uint32_t jm, int L, int shift,
                               uint32_t max_q_resident, uint32_t* d_items, unsigned long long* d_stats,
                               uint32_t* d_nslices, const uint32_t* d_seg_of, hipStream_t s) {
  hs_seg_route_kernel<<<blocks_for((uint64_t)n_max + 1), 256, 0, s>>>(
      d_seg_key, d_seg_cnt, d_seg_qoff, d_n_seg, d_sorted_ql, d_qcount, n_max, min_q, min_m, jm, L,
      shift, max_q_resident, d_items, d_stats);
  // (d_nslices == null: every segment with a member goes to the join -- min_q = min_m = 1 -- and the caller
  // clears the slice counts of ALL probes with one memset instead of this kernel's scattered stores)
  if (d_nslices)
    hs_seg_unslice_kernel<<<blocks_for(n_max), 256, 0, s>>>(d_seg_of, d_items, d_sorted_ql, n_max, d_nslices);
  return hipGetLastError();
}

hipError_t hs_launch_seg_big(const uint32_t* d_seg_cnt, const uint32_t* d_items, uint32_t n,
                             uint32_t min_q, uint32_t max_q_resident, uint32_t* d_big, uint32_t* d_res,
                             hipStream_t s) {
  hs_seg_big_kernel<<<blocks_for(n), 256, 0, s>>>(d_seg_cnt, d_items, n, min_q, max_q_resident, d_big, d_res);
  return hipGetLastError();
}
hipError_t hs_launch_seg_order(const uint32_t* d_big_pos, const uint32_t* d_res_pos, const uint32_t* d_items,
                               uint32_t n, uint32_t* d_order, uint32_t* d_items_ordered, hipStream_t s) {
  hs_seg_order_kernel<<<blocks_for(n), 256, 0, s>>>(d_big_pos, d_res_pos, d_items, n, d_order, d_items_ordered);
  return hipGetLastError();
}
hipError_t hs_launch_item_split(const uint32_t* d_item_off, const uint32_t* d_res_pos, uint32_t n,
                                uint32_t* d_split, hipStream_t s) {
  hs_item_split_kernel<<<1, 64, 0, s>>>(d_item_off, d_res_pos, n, d_split);
  return hipGetLastError();
}

hipError_t hs_launch_item_desc(const hs_tables_dev& tabs, const uint64_t* d_seg_key,
                               const uint32_t* d_seg_cnt,
                               const uint32_t* d_seg_qoff, const uint32_t* d_item_off, uint32_t n_max,
                               const uint32_t* d_sorted_ql, const uint32_t* d_qcount, uint32_t n_items,
                               uint32_t jm, int shift, const uint32_t* d_order, int PW,
                               const uint32_t* d_n_items, uint4* d_desc, hipStream_t s) {
  if (!n_items) return hipSuccess;
  hs_item_desc_kernel<<<blocks_for(n_items), 256, 0, s>>>(tabs, d_seg_key, d_seg_cnt, d_seg_qoff, d_item_off,
                                                          n_max, d_sorted_ql, d_qcount, n_items, jm, shift,
                                                          d_order, PW, d_n_items, d_desc);
  return hipGetLastError();
}

hipError_t hs_launch_gather_c16(const void* d_c16, const uint32_t* d_sorted_ql, uint32_t nql, int L,
                                void* d_out, hipStream_t s) {
  if (!nql) return hipSuccess;
  hs_gather_c16_kernel<<<blocks_for((uint64_t)nql * JG), 256, 0, s>>>(
      (const _Float16*)d_c16, d_sorted_ql, nql, L, (_Float16*)d_out);
  return hipGetLastError();
}

hipError_t hs_launch_join(const uint4* d_desc, uint32_t n_items, const uint4* d_packed_base,
                          const uint32_t* d_sorted_ql, const void* d_c16s, const void* d_tab16,
                          const float* d_rownorm, int k, uint32_t* d_prov_count, uint32_t prov_cap,
                          uint2* d_prov, int n_blocks, hipStream_t s) {
  if (!n_items) return hipSuccess;
  hs_join_kernel<<<n_blocks, 256, 0, s>>>(d_desc, n_items, d_packed_base, d_sorted_ql, (const _Float16*)d_c16s,
                                          (const _Float16*)d_tab16, d_rownorm, k, d_prov_count,
                                          prov_cap, d_prov);
  return hipGetLastError();
}
