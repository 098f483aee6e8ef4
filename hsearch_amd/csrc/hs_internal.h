// hs_internal.h -- declarations shared by the translation units of libhsearch_amd.so.
// gfx950 (MI355X) only; no CPU fallback anywhere in this directory.
#ifndef HS_INTERNAL_H
#define HS_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hsearch.h"

#define HS_MAX_L 32
#define HS_MAX_K 32
#define HS_ALPHABET_PAD 32  // table rows addressable by a 5-bit residue code
#define HS_TROW 32          // floats per row of a per-query distance table (20 used)
// survivor entries {probe, entry position}: probe = q_local * L + table, or -- from the join
// kernels -- HS_PROV_INDIRECT | position of the probe in segment order (index into sorted_ql)
#define HS_PROV_INDIRECT 0x80000000u
#define HS_SLICE 4096u      // candidates one wavefront scans per work item
#define HS_KEY_CHARS (11 * HS_MAX_K + 1)

// ---- key fingerprints ---------------------------------------------------------------------------
// The reference keys a table by the STRING to_string(b_0)+...+to_string(b_{K-1}) (lsh.hpp:51-59).
// The index keys by a 64-bit fingerprint of exactly that character stream (so tuples whose strings
// alias, e.g. (1,23) and (12,3), share a fingerprint by construction) and verifies string equality
// exactly: at build time for every sorted neighbour pair, at probe time against the bucket's tuple.
__host__ __device__ inline uint64_t hs_key_init(uint32_t seed) {
  return 0xcbf29ce484222325ull ^ ((uint64_t)(seed + 1) * 0x9e3779b97f4a7c15ull);
}
__host__ __device__ inline uint64_t hs_key_put(uint64_t h, uint32_t ch) {
  return (h ^ ch) * 0x100000001b3ull;
}
__host__ __device__ inline uint64_t hs_key_put_int(uint64_t h, int32_t v) {
  // the decimal characters of v, most significant first (std::to_string, lsh.hpp:51-59); divisions
  // by constants only, short numbers (the usual bucket ints) first
  uint32_t m;
  if (v < 0) {
    h = hs_key_put(h, '-');
    m = 0u - (uint32_t)v;
  } else {
    m = (uint32_t)v;
  }
  if (m < 10u) return hs_key_put(h, '0' + m);
  if (m < 100u) {
    const uint32_t q = m / 10u;
    h = hs_key_put(h, '0' + q);
    return hs_key_put(h, '0' + (m - 10u * q));
  }
  bool started = false;
#define HS_DIGIT(P)                            \
  {                                            \
    const uint32_t dgt = (m / (P)) % 10u;      \
    started = started || dgt != 0u;            \
    if (started) h = hs_key_put(h, '0' + dgt); \
  }
  HS_DIGIT(1000000000u) HS_DIGIT(100000000u) HS_DIGIT(10000000u) HS_DIGIT(1000000u) HS_DIGIT(100000u)
  HS_DIGIT(10000u) HS_DIGIT(1000u) HS_DIGIT(100u) HS_DIGIT(10u) HS_DIGIT(1u)
#undef HS_DIGIT
  return h;
}
__host__ __device__ inline uint64_t hs_key_fin(uint64_t h) {
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ull;
  h ^= h >> 33;
  return h;
}
__host__ __device__ inline uint64_t hs_key_of(const int32_t* t, int K, uint32_t seed) {
  uint64_t h = hs_key_init(seed);
  for (int i = 0; i < K; ++i) h = hs_key_put_int(h, t[i]);
  return hs_key_fin(h);
}
// Decimal characters of the concatenation; returns the length.
__host__ __device__ inline int hs_key_chars(const int32_t* t, int K, char* out) {
  int n = 0;
  for (int i = 0; i < K; ++i) {
    int32_t v = t[i];
    uint32_t m;
    if (v < 0) {
      out[n++] = '-';
      m = 0u - (uint32_t)v;
    } else {
      m = (uint32_t)v;
    }
    uint32_t p = 1;
    while (m / p >= 10) p *= 10;
    while (p) {
      uint32_t dgt = m / p;
      out[n++] = (char)('0' + dgt);
      m -= dgt * p;
      p /= 10;
    }
  }
  return n;
}
// HashKey string equality of two K-tuples (fast path: identical tuples).
__host__ __device__ inline bool hs_key_equal(const int32_t* x, const int32_t* y, int K) {
  bool same = true;
  for (int i = 0; i < K; ++i) same = same && (x[i] == y[i]);
  if (same) return true;
  char sx[HS_KEY_CHARS], sy[HS_KEY_CHARS];
  int nx = hs_key_chars(x, K, sx), ny = hs_key_chars(y, K, sy);
  if (nx != ny) return false;
  for (int i = 0; i < nx; ++i)
    if (sx[i] != sy[i]) return false;
  return true;
}

// ---- device view of one hash table --------------------------------------------------------------
// The survivor list's counter is 32 bits (the batch's counters block, word 0).  A batch whose filters
// pass more than ~4e9 pairs (a radius close to the typical distance of bucket mates) would wrap it
// silently: every reservation that lands in the last 2^28 slots raises HS_CNT_SURVIVOR_OVERFLOW in
// the same block -- reservations are <= 256 slots (hs_join8.hip JRES), so one does before the counter
// wraps -- and the host repeats the batch in halves (hs_capi.hip run_query).
#define HS_CNT_SURVIVOR_OVERFLOW 21
// hs_query_codes: a query's residue code lay outside the alphabet (hs_check_codes_kernel)
#define HS_CNT_BAD_QUERY_CODE 22
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t hs_reserve_survivors(uint32_t* prov_count, uint32_t n) {
  const uint32_t base = atomicAdd(prov_count, n);
  if (base >= 0xF0000000u) atomicOr(prov_count + HS_CNT_SURVIVOR_OVERFLOW, 1u);
  return base;
}
#endif

struct hs_table_dev {
  const uint64_t* dir_key;   // [nb] sorted fingerprints of the distinct keys
  const uint32_t* dir_start; // [nb+1] first sorted position of each bucket
  const int32_t* dir_tuple;  // [nb][K] bucket ints of the bucket's first member
  const uint4* packed;       // [n][PW] 5-bit packed residue codes in bucket order
  const uint32_t* ids;       // [n] DB ids in bucket order (ascending inside a bucket)
  const uint32_t* pos_of;    // [n] inverse of ids: sorted position of DB id i in this table
  const uint32_t* dir_jump;  // [2^J + 1] first directory entry whose fingerprint's top J bits are >= the slot
  const uint64_t* giant_key; // [n_giant] ascending TUPLE HASHES (hs_tuple_hash of the bucket ints) of the buckets with
  uint32_t n_giant;          // more than hs_giant_threshold members (bucket partition: shared among the parts by QUERY)
  const uint4* dir_rec;      // [nb][4] or null: one 64-byte line per bucket with all a probe reads of it --
                             // {fingerprint, start, count} and the K <= HS_REC_MAX_K bucket ints as int16
                             // (null: K larger, an int outside 16 bits, or the option is off)
  uint32_t nb;
  uint32_t jump_shift;       // 64 - J
};
struct hs_tables_dev {
  hs_table_dev t[HS_MAX_L];
  // bucket partition (hs_set_bucket_partition): the probe kernel looks only for the (query, bucket) probes
  // that fall to `part` of `n_parts` (hs_probe_part); n_parts <= 1: all of them.  q_first: number, in the
  // call, of the batch's first query.  probe_list != null: the kernel probes these n_list (query, table) pairs
  // only -- the part's own, found ahead of it (hs_launch_part_owned).
  uint32_t part, n_parts, q_first;
  const uint32_t* probe_list;
  uint32_t n_list;
};
// A cheap hash of a probe's K bucket ints (t[0], t[stride], ...): what decides the part of a probe.  (Not the
// key fingerprint: that one walks the decimal characters of every int, ~ 10 x the instructions, and seven
// probes in eight of a batch belong to other parts.  Tuples whose key STRINGS coincide may fall to different
// parts -- each probe still belongs to exactly one.)
__host__ __device__ inline uint64_t hs_tuple_hash(const int32_t* t, int K, int stride) {
  uint64_t h = 0x243f6a8885a308d3ull;
  for (int j = 0; j < K; ++j) {
    h = (h ^ (uint32_t)t[(size_t)j * stride]) * 0x9e3779b97f4a7c15ull;
    h ^= h >> 29;
  }
  return h;
}
// The part a probe belongs to, of n_parts: a function of its bucket ints -- every rank the same answer, the
// parts even whatever the tables look like -- and, for the few GIANT buckets alone, of the query's number in
// the call as well: a table's largest buckets each hold per cents of all (member, query) pairs (configs[2]:
// 600 buckets of > 10^5 members hold half of them), and whole they would land on the parts like rocks; shared
// by query every part gets its n-th of each.
__host__ __device__ inline uint32_t hs_probe_part(uint64_t tuple_hash, bool giant, uint32_t q, uint32_t n_parts) {
  if (giant) tuple_hash ^= (uint64_t)(q + 1u) * 0x9e3779b97f4a7c15ull;
  return (uint32_t)(((tuple_hash >> 20) & 0xffffffffull) % n_parts);
}
// more members than this make a bucket a giant (n = k-mers of the index)
static inline uint32_t hs_giant_threshold(uint64_t n) { return (uint32_t)(n / 1024 > 4096 ? n / 1024 : 4096); }

// 16-byte words per packed k-mer
// 25 residues x 5 bits per 16-byte word
static inline int hs_packed_words(int k) { return (k + 24) / 25; }

// ---- primitive wrappers (hs_prims.hip, rocPRIM behind them) --------------------------------------
size_t hs_sort_pairs_u64_u32_temp(size_t n);
// keys ordered by their bits [begin_bit, end_bit) only; begin_bit > 0 is refused (hipErrorInvalidValue)
// unless hs_sort_partial_bits_ok(n): only rocPRIM's onesweep path handles such a range (hs_prims.hip)
bool hs_sort_partial_bits_ok(size_t n);
hipError_t hs_sort_pairs_u64_u32(void* temp, size_t temp_bytes, const uint64_t* kin, uint64_t* kout,
                                 const uint32_t* vin, uint32_t* vout, size_t n, int begin_bit, int end_bit,
                                 hipStream_t s);
size_t hs_sort_pairs_u32_u32_temp(size_t n);
hipError_t hs_sort_pairs_u32_u32(void* temp, size_t temp_bytes, const uint32_t* kin, uint32_t* kout,
                                 const uint32_t* vin, uint32_t* vout, size_t n, int end_bit, hipStream_t s);
size_t hs_sort_pairs_u64_u64_temp(size_t n);
hipError_t hs_sort_pairs_u64_u64(void* temp, size_t temp_bytes, const uint64_t* kin, uint64_t* kout,
                                 const uint64_t* vin, uint64_t* vout, size_t n, int end_bit,
                                 hipStream_t s);
size_t hs_scan_u32_temp(size_t n);
hipError_t hs_exclusive_scan_u32(void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out,
                                 size_t n, hipStream_t s);
size_t hs_rle_u64_temp(size_t n);
// unique_out[n], counts_out[n], runs_out[1] (device)
hipError_t hs_rle_u64(void* temp, size_t temp_bytes, const uint64_t* in, uint64_t* unique_out,
                      uint32_t* counts_out, uint32_t* runs_out, size_t n, hipStream_t s);

// ---- grouping a table's k-mers by key without sorting fingerprints (hs_group.hip) -----------------
// bucket ints -> slots of an open-addressing table d_table[C] of 64-bit fingerprints (C =
// hs_group_table_slots(n) slots): d_slot_of[i] = slot of k-mer i's key.  *d_flag |= 16: the table filled up (or
// a fingerprint equals its empty marker) -- the caller takes the sorting path for this table.
// hs_launch_group_check, once the buckets have their tuples: the exact-membership proof of every k-mer (its
// bucket ints against the tuple of the bucket of rank d_rank_of[i]); *d_flag |= 1: one fingerprint, two HashKey
// strings (rebuild with another seed).
uint32_t hs_group_table_slots(uint64_t n);
hipError_t hs_launch_iota_u32(uint32_t* d_out, uint32_t n, hipStream_t s);
hipError_t hs_launch_group_insert(const int32_t* d_ints, uint64_t n, int K, uint32_t seed, uint64_t* d_table,
                                  uint32_t C, uint32_t* d_slot_of, uint32_t* d_flag, hipStream_t s);
hipError_t hs_launch_group_check(const int32_t* d_ints, uint64_t n, int K, const uint32_t* d_rank_of,
                                 const int32_t* d_dir_tuple, uint32_t* d_flag, hipStream_t s);
// distinct keys per block of 1024 slots; after an exclusive scan of those counts (d_blk_off, with the
// total at [n_blocks]): the distinct keys' fingerprints d_dk and their slots d_ds in slot order
hipError_t hs_launch_fp_count(const uint64_t* d_table, uint32_t C, uint32_t* d_blk_cnt, hipStream_t s);
hipError_t hs_launch_fp_compact(const uint64_t* d_table, uint32_t C, const uint32_t* d_blk_off, uint64_t* d_dk,
                                uint32_t* d_ds, hipStream_t s);
// d_rank_of_slot[d_ds_sorted[r]] = r; then d_slot_of[i] <- rank of k-mer i's key, in place
hipError_t hs_launch_rank_slots(const uint32_t* d_ds_sorted, uint32_t nb, uint32_t* d_rank_of_slot, hipStream_t s);
hipError_t hs_launch_rank_kmers(uint32_t* d_slot_of, uint64_t n, const uint32_t* d_rank_of_slot, hipStream_t s);
// one stable LSD radix pass (8 bits from `shift`) over (key, id): histogram [256][hs_rs_blocks(n)], the
// caller's exclusive scan over it, scatter (d_ids_in null: ids 0 .. n - 1)
uint32_t hs_rs_blocks(uint64_t n);
hipError_t hs_launch_rs_hist(const uint32_t* d_keys, uint32_t n, int shift, uint32_t* d_hist, hipStream_t s);
hipError_t hs_launch_rs_scatter(const uint32_t* d_keys_in, const uint32_t* d_ids_in, uint32_t n, int shift,
                                const uint32_t* d_hist_scanned, uint32_t* d_keys_out, uint32_t* d_ids_out,
                                hipStream_t s);
// bucket boundaries from the sorted ranks (d_dir_start[nb + 1]) and the largest bucket (atomicMax into *d_max)
hipError_t hs_launch_dir_start(const uint32_t* d_ranks_sorted, uint32_t n, uint32_t nb, uint32_t* d_dir_start,
                               uint32_t* d_max, hipStream_t s);

// index build with the hashing spread over ranks: d_tuples[nb][K] = the bucket ints of every bucket whose
// first member lies in this rank's block [lo, lo + cnt) (zeros elsewhere); the exact-membership proof of
// the block's k-mers against the buckets' tuples (*d_flag |= 1: one fingerprint, two HashKey strings)
hipError_t hs_launch_shard_first_tuples(const uint32_t* d_dir_start, const uint32_t* d_ids, const int32_t* d_ints_block,
                                        uint32_t lo, uint32_t cnt, uint32_t nb, int K, int32_t* d_tuples,
                                        hipStream_t s);
hipError_t hs_launch_shard_check(const int32_t* d_ints_block, uint32_t lo, uint32_t cnt, int K,
                                 const uint32_t* d_pos_of, const uint32_t* d_dir_start, uint32_t nb,
                                 const int32_t* d_dir_tuple, uint32_t* d_flag, hipStream_t s);

// ---- projection on the matrix cores (hs_proj.hip) ------------------------------------------------
// The coordinate table in 16-bit fixed point for the codes path: per residue the high and low digit
// bytes of its 8 coordinates, an upper bound of its 1-norm, the scale 2^-ex and dx = 2^-(ex+1).
struct hs_proj_table {
  uint4 dig[HS_ALPHABET_PAD];   // {X1[0..3], X1[4..7], X0[0..3], X0[4..7]}
  double l1[HS_ALPHABET_PAD];
  double sx, dx, l1max;
  uint32_t unsafe, pad;
};
// k-steps (of 32 dimensions) the MFMA pass is compiled for: 4, 7, 10 or 13; 0 = k too long (> 52)
int hs_proj_steps(int k);
hipError_t hs_launch_quant_table(const double* d_coords, int alphabet, hs_proj_table* d_tab, hipStream_t s);
// planes -> digit fragments for the all-functions tiling (d_aq_all: ceil(F/32) tiles) and the
// per-table tiling (d_aq_tab: one tile per table; both zero-filled by the caller, S*2*64 uint4 per
// tile) + constants d_fn[F] (4 doubles each); d_stats[3] (zeroed): max da, max |a^|_1 as double
// bits, and 1 if a function cannot be quantised
hipError_t hs_launch_quant_planes(const double* d_a, const double* d_b, int F, int d, int K, int S, double W,
                                  double eps_scale, void* d_aq_all, void* d_aq_tab, void* d_fn,
                                  unsigned long long* d_stats, hipStream_t s);
// points -> d_xq [n][S*4] uint4 + d_xmeta [n][3]
hipError_t hs_launch_quant_points(const double* d_pts, uint64_t n, int k, int S, void* d_xq, double* d_xmeta,
                                  hipStream_t s);
// fast pass: bucket ints of F functions (d_aq / d_fn already offset to the first of them) for n
// points given as codes (d_codes != null) or as quantised points; uncertain values are appended to
// d_flags (*d_flag_count zeroed by the caller) ...
hipError_t hs_launch_proj(const uint8_t* d_codes, const void* d_xq, const double* d_xmeta, uint64_t n, int k,
                          int S, const void* d_aq, const void* d_fn, int F, const hs_proj_table* d_tab,
                          double W, int32_t* d_out, int out_stride, uint2* d_flags, uint32_t flag_cap,
                          uint32_t* d_flag_count, int n_cu, hipStream_t s);
// ... and recomputed here in the reference's operation order (everything, if the list overflowed)
hipError_t hs_launch_proj_fix(const uint8_t* d_codes, const double* d_pts, uint64_t n, int k, const double* d_aT,
                              int ldf, const double* d_b, int F, double W, const double* d_coords,
                              int32_t* d_out, int out_stride, const uint2* d_flags, uint32_t flag_cap,
                              const uint32_t* d_flag_count, hipStream_t s);

// ---- kernel launchers (hs_kernels.hip) -----------------------------------------------------------
hipError_t hs_launch_embed(const uint8_t* d_codes, uint64_t n, int k, const double* d_coords,
                           double* d_out, hipStream_t s);
// buckets out[i*out_stride + f], f in [0,F): exact reference arithmetic.  d_aT = the plane matrix
// TRANSPOSED, [8k][ldf] doubles (dimension-major), already offset to the first of the F functions.
hipError_t hs_launch_hash_codes(const uint8_t* d_codes, uint64_t n, int k, const double* d_aT, int ldf,
                                const double* d_b, int F, double W, const double* d_coords,
                                int32_t* d_out, int out_stride, hipStream_t s);
hipError_t hs_launch_hash_points(const double* d_pts, uint64_t n, int k, const double* d_aT, int ldf,
                                 const double* d_b, int F, double W, int32_t* d_out, int out_stride,
                                 hipStream_t s);
// d_out[c][r] = d_in[r][c]
hipError_t hs_launch_transpose_f64(const double* d_in, int rows, int cols, double* d_out, hipStream_t s);
// keys[i] = fingerprint(ints[i*stride .. +K)); ids[i] = i (if ids != null)
hipError_t hs_launch_keys(const int32_t* d_ints, uint64_t n, int stride, int K, uint32_t seed,
                          uint64_t* d_keys, uint32_t* d_ids, hipStream_t s);
// flag |= 1 if two sorted neighbours share a fingerprint but not a key string; flag |= 2 if the
// queue of non-identical neighbour tuples (d_slow: count + slow_cap positions) overflowed, in which
// case the caller repeats the call with exhaustive = true (every neighbour pair compared as strings)
hipError_t hs_launch_check_runs(const uint64_t* d_keys_sorted, const uint32_t* d_ids_sorted,
                                const int32_t* d_ints, uint64_t n, int K, uint32_t* d_flag,
                                uint32_t* d_slow, uint32_t slow_cap, bool exhaustive, int sorted_from_bit,
                                hipStream_t s);
hipError_t hs_launch_dir_tuples(const uint32_t* d_dir_start, const uint32_t* d_ids_sorted,
                                const int32_t* d_ints, uint32_t nb, int K, int32_t* d_dir_tuple,
                                hipStream_t s);
hipError_t hs_launch_pack(const uint8_t* d_codes, uint64_t n, int k, int alphabet, uint4* d_packed,
                          uint32_t* d_bad, hipStream_t s);
hipError_t hs_launch_gather_packed(const uint4* d_packed_all, const uint32_t* d_ids_sorted,
                                   uint64_t n, int PW, uint4* d_out, hipStream_t s);
hipError_t hs_launch_set_u32(uint32_t* d_p, uint32_t v, hipStream_t s);
// d_out[i] = d_in[i] if it is a row of the table (< alphabet), else 0 and *d_bad |= 1
hipError_t hs_launch_recognise_kmers(const double* d_centers, uint64_t nq, int k, const double* d_coords, int alphabet,
                                     uint8_t* d_out_codes, uint32_t* d_n_unrecognised, hipStream_t s);
hipError_t hs_launch_check_codes(const uint8_t* d_in, uint64_t n_bytes, int alphabet, uint8_t* d_out,
                                 uint32_t* d_bad, hipStream_t s);
// windows of length k of every sequence of a residue buffer -> codes [n_windows][k] (+ the buffer
// position of every window); d_win_off[s] = number of the first window of sequence s
hipError_t hs_launch_windows(const uint8_t* d_residues, uint32_t n_residues, const uint32_t* d_seq_start,
                             const uint32_t* d_win_off, uint32_t n_seq, int k, uint8_t* d_codes,
                             uint32_t* d_win_pos, hipStream_t s);
// KLSH codes of n_seq sequences of reduced-alphabet classes (one wave per sequence)
hipError_t hs_launch_klsh(const uint8_t* d_classes, const uint64_t* d_seq_start, uint64_t n_seq,
                          const double* d_w, const double* d_b, const double* d_t, uint32_t bits,
                          uint64_t* d_codes, uint64_t* d_uncertain, hipStream_t s);
// d_out[d_perm[i]] = i
hipError_t hs_launch_invert_perm(const uint32_t* d_perm, uint32_t n, uint32_t* d_out, hipStream_t s);
hipError_t hs_launch_gather_rows(const uint8_t* d_all, const uint32_t* d_subset, uint64_t n_sub, int k,
                                 uint8_t* d_out, hipStream_t s);
// hs_index_load: checks of a table read from a file (see hs_validate_*_kernel); writes the inverse
// permutation into d_pos_of, ORs failure bits into *d_flag, atomicMax of the bucket sizes into
// *d_max_bucket
hipError_t hs_launch_validate_table(const uint32_t* d_ids, uint32_t n, uint32_t* d_pos_of,
                                    const uint32_t* d_dir_start, const uint64_t* d_dir_key,
                                    const int32_t* d_dir_tuple, uint32_t nb, int K, uint32_t seed,
                                    uint32_t* d_flag, uint32_t* d_max_bucket, hipStream_t s);
// the tuple hashes of the buckets with more than `threshold` members, unordered: d_out[atomicAdd(d_count, 1)]
// while the count stays below cap (the count keeps running: the caller sees an overflow)
hipError_t hs_launch_giant_buckets(const int32_t* d_dir_tuple, int K, const uint32_t* d_dir_start, uint32_t nb,
                                   uint32_t threshold, uint64_t* d_out, uint32_t cap, uint32_t* d_count, hipStream_t s);
// Bucket partition, ahead of the probe: d_flag[ql] = 1 where probe ql = (query, table) belongs to tabs.part (by
// its bucket ints alone: no fingerprint, no directory); the outputs of the probe kernel are cleared for the
// others (d_qbucket[ql] = nb_total: found nothing).  d_flag[nq * L] = 0.
hipError_t hs_launch_part_owned(const hs_tables_dev& tabs, const int32_t* d_qints, uint32_t nq, int K, int L,
                                uint32_t nb_total, uint32_t* d_flag, uint32_t* d_qstart, uint32_t* d_qcount,
                                uint32_t* d_nslices, uint64_t* d_cand_out, uint32_t* d_qbucket, hipStream_t s);
// d_list[d_pos[i]] = i where d_flag[i] (d_pos = exclusive scan of d_flag)
hipError_t hs_launch_flagged_list(const uint32_t* d_flag, const uint32_t* d_pos, uint32_t n, uint32_t* d_list,
                                  hipStream_t s);
#define HS_REC_MAX_K 24
// rec[b] = {key[b], start[b], start[b + 1] - start[b]; int16 tuple[b][0..K)}; *d_flag |= 1 where an int does not fit
hipError_t hs_launch_dir_records(const uint64_t* d_dir_key, const uint32_t* d_dir_start, const int32_t* d_dir_tuple,
                                 uint32_t nb, int K, uint4* d_rec, uint32_t* d_flag, hipStream_t s);
// jump[t] = first directory entry with (key >> shift) >= t, t = 0 .. n_slots (jump[n_slots] = nb)
hipError_t hs_launch_dir_jump(const uint64_t* d_dir_key, uint32_t nb, uint32_t shift, uint32_t n_slots,
                              uint32_t* d_jump, hipStream_t s);
hipError_t hs_launch_max_u32(const uint32_t* d_in, uint32_t n, uint32_t* d_out, hipStream_t s);

hipError_t hs_launch_probe(const hs_tables_dev& tabs, const int32_t* d_qints, uint32_t nq, int K,
                           int L, uint32_t seed, uint32_t* d_qstart, uint32_t* d_qcount,
                           uint32_t* d_nslices, uint64_t* d_cand_out, unsigned long long* d_cand_total,
                           uint32_t* d_slow /* [nq*L + 1] */, const uint32_t* d_dir_base,
                           uint32_t nb_total, uint32_t* d_bucket_count, uint32_t* d_qbucket,
                           uint32_t* d_qrank, hipStream_t s);
// With d_bucket_count != null the probe also groups the probes by bucket for the join: global
// bucket number d_qbucket[ql] = d_dir_base[table] + directory index (nb_total = the pseudo-bucket
// of probes that found none) and d_qrank[ql] = arrival rank inside it (d_bucket_count[nb_total + 2]).
// hs_launch_seg_group turns that into sorted_ql / seg_key / seg_cnt / n_seg (a counting sort: the
// segments come out in bucket-number order, the pseudo-bucket last).
// The same outputs when buckets far outnumber probes (the counting sort's passes over every bucket
// slot would dominate: C3 shape at W = 160, 1.3e8 slots for 4e6 probes): the probes are radix-sorted
// on their bucket number (d_qbucket from the probe kernels, which then take no ranks), segment
// heads found by comparing neighbours.  d_work: 4 (nql + 1) words; d_iota / d_keys_sorted: nql words.
hipError_t hs_launch_seg_group_sparse(const hs_tables_dev& tabs, const uint32_t* d_dir_base, int L, int shift,
                                      uint32_t nb_total, void* d_temp, size_t temp_bytes,
                                      const uint32_t* d_qbucket, uint32_t* d_keys_sorted, uint32_t* d_iota,
                                      uint32_t* d_work, uint32_t nql, uint32_t* d_sorted_ql,
                                      uint64_t* d_seg_key, uint32_t* d_seg_cnt, uint32_t* d_n_seg,
                                      uint32_t* d_seg_of /* [nql]: segment of every sorted probe position */,
                                      hipStream_t s, const uint32_t* d_probes_in = nullptr);
// bucket partition: the probes that found a bucket (d_qbucket[ql] != nb_total), in probe order, as (bucket,
// probe) pairs d_keys / d_probes; d_pos[nql] = their number.  d_flag, d_pos: nql + 1 words.  d_list != null: only
// the nql probes d_list[0 .. nql) (ascending) are looked at.
hipError_t hs_launch_found_probes(const uint32_t* d_qbucket, uint32_t nql, uint32_t nb_total, void* d_temp,
                                  size_t temp_bytes, uint32_t* d_flag, uint32_t* d_pos, uint32_t* d_keys,
                                  uint32_t* d_probes, hipStream_t s, const uint32_t* d_list = nullptr);
hipError_t hs_launch_seg_group(const hs_tables_dev& tabs, const uint32_t* d_dir_base, int L, int shift,
                               uint32_t nb_total, const uint32_t* d_bucket_count,
                               uint32_t* d_bucket_work /* 3 x (nb_total + 2) */, void* d_temp,
                               size_t temp_bytes, const uint32_t* d_qbucket, const uint32_t* d_qrank,
                               uint32_t nql, uint32_t* d_sorted_ql, uint64_t* d_seg_key,
                               uint32_t* d_seg_cnt, uint32_t* d_n_seg, uint32_t* d_seg_of, hipStream_t s);
// self-join: query q = indexed k-mer first_id + q probes the bucket it sits in (no hash, no directory
// search); outputs as hs_launch_probe
hipError_t hs_launch_self_probe(const hs_tables_dev& tabs, uint32_t first_id, uint32_t nq, int L,
                                uint32_t* d_qstart, uint32_t* d_qcount, uint32_t* d_nslices,
                                uint64_t* d_cand_out, unsigned long long* d_cand_total,
                                const uint32_t* d_dir_base, uint32_t nb_total, uint32_t* d_bucket_count,
                                uint32_t* d_qbucket, uint32_t* d_qrank, hipStream_t s);
hipError_t hs_launch_qtables(const double* d_centers, uint32_t nq, int k, const double* d_coords,
                             int alphabet, float* d_tq, hipStream_t s);
hipError_t hs_launch_verify(const hs_tables_dev& tabs, const uint32_t* d_qstart,
                            const uint32_t* d_qcount, const uint32_t* d_slice_off, uint32_t nql,
                            const float* d_tq, int k, int L, float r2_hi, uint32_t* d_prov_count,
                            uint32_t prov_cap, uint2* d_prov, int n_blocks, hipStream_t s);
// d_qcodes != null: the queries are indexed k-mers given as codes [nq][k] (self-join), d_centers unused
hipError_t hs_launch_finalize(const hs_tables_dev& tabs, const uint8_t* d_codes,
                              const double* d_centers, const uint8_t* d_qcodes, const double* d_coords,
                              const uint32_t* d_qstart, const uint32_t* d_qcount,
                              const uint2* d_prov, const uint32_t* d_prov_count, uint32_t prov_cap,
                              const uint32_t* d_sorted_ql, int k, int L, double r2, double r_sqrt,
                              uint32_t q_base, uint32_t self_first, uint32_t* d_hit_count,
                              uint32_t hit_cap, uint64_t* d_hit_key, uint64_t* d_hit_val,
                              uint32_t* d_qcnt /* [nq] hits per query, or null */, int alphabet,
                              const uint4* d_qpacked /* the queries as packed k-mers (with d_qcodes), or null */,
                              uint32_t* d_hit_rank /* with d_qcnt: the hit's number among its query's hits */,
                              hipStream_t s);
#ifdef __HIPCC__
// First-seen rule (label[], motif_both_points.cpp:233): a hit's id was already reported if an EARLIER
// table's probed bucket holds it, i.e. if its sorted position in that table falls inside the bucket's
// range.  These kernels are bound by the ADDRESSES their scattered loads present (a wave instruction with
// 64 different cache lines occupies the address unit for 64 cycles), so: four tables per step; their four
// (start, count) words as two 16-byte loads (qstart / qcount carry four words of padding for the last
// query); one position load per table, each under the lanes that need that table only.
__device__ __forceinline__ bool seen_in_earlier_table(const hs_tables_dev& tabs, const uint32_t* __restrict__ qstart,
                                                      const uint32_t* __restrict__ qcount, uint32_t q, int l, int L,
                                                      uint32_t id, bool hit) {
  bool dup = false;
  struct __attribute__((packed, aligned(4))) U4 { uint32_t v[4]; };
  // steps of 1, 3, 4, 4, ... tables: a pair that shares a bucket in one table mostly shares table 0's too, and
  // a hit found there needs no further look (every look is a cache line of pos_of from HBM)
  int w = 1;
  for (int l0 = 0; l0 < L; l0 += w, w = l0 == 1 ? 3 : 4) {
    if (!__ballot(hit && l > l0 && !dup)) break;  // (wave-uniform)
    if (hit && l > l0 && !dup) {
      const U4 c4 = *reinterpret_cast<const U4*>(qcount + (size_t)q * L + l0);
      const U4 s4 = *reinterpret_cast<const U4*>(qstart + (size_t)q * L + l0);
      uint32_t p4[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u < w && l0 + u < l) p4[u] = tabs.t[l0 + u].pos_of[id];
#pragma unroll
      for (int u = 0; u < 4; ++u) dup = dup || (u < w && l0 + u < l && p4[u] - s4.v[u] < c4.v[u]);
    }
  }
  return dup;
}
#endif

// hs_finalize's first-seen test reads four words at a time from d_qstart / d_qcount: this many words of padding
#define HS_QRANGE_PAD 4
// the batch's hits in the reference's order without a sort: bucket by query (d_qoff = exclusive
// scan of the per-query counts), order every query's few hits, unpack to the outputs (at most
// out_room of them); *d_big is set when a query has too many hits for that (caller: radix sort)
hipError_t hs_launch_hit_order(const uint64_t* d_key, const uint64_t* d_val, const uint32_t* d_hit_count,
                               uint32_t hit_cap, uint32_t q_base, uint32_t nq, const uint32_t* d_qoff,
                               const uint32_t* d_rank, void* d_kv /* hit_cap x 16 bytes: (key, value) by query */, uint32_t* d_big,
                               uint32_t* d_qlist /* 8 + 3 nq words, the first eight zero */,
                               uint32_t* d_q, uint32_t* d_id, uint32_t* d_table, double* d_dist,
                               uint64_t out_room, int n_cu, hipStream_t s);
// self_first: the queries are the indexed k-mers self_first, self_first + 1, ... themselves (the
// self-join): the pair of a k-mer with itself is not a hit; HS_NO_SELF otherwise
#define HS_NO_SELF 0xffffffffu
// merge of the table-partitioned layout (hs_merge_first_table_dev): keys (q, id, table) + the distance bits;
// run heads of the sorted keys; the heads re-keyed (q, table, id) at their scanned positions
hipError_t hs_launch_merge_key1(const uint32_t* d_q, const uint32_t* d_id, const uint32_t* d_table, const double* d_dist,
                                uint32_t n, uint64_t* d_key, uint64_t* d_val, hipStream_t s);
hipError_t hs_launch_merge_flag(const uint64_t* d_key, uint32_t n, uint32_t* d_flag /* [n + 1] */, hipStream_t s);
hipError_t hs_launch_merge_compact(const uint64_t* d_key, const uint64_t* d_val, const uint32_t* d_pos /* [n + 1] */,
                                   uint32_t n, uint64_t* d_key2, uint64_t* d_val2, hipStream_t s);
hipError_t hs_launch_unpack_hits(const uint64_t* d_key, const uint64_t* d_val, uint32_t n,
                                 uint32_t* d_q, uint32_t* d_id, uint32_t* d_table, double* d_dist,
                                 hipStream_t s);
// brute force
hipError_t hs_launch_bruteforce(const uint4* d_packed_all, uint32_t n, const float* d_tq,
                                uint32_t nq, int k, float r2_hi, uint32_t* d_prov_count,
                                uint32_t prov_cap, uint2* d_prov, const float* d_q_thr,
                                float* d_slice_min, int n_blocks, hipStream_t s);
// bucket join (hs_join.hip)
hipError_t hs_launch_jtables(const double* d_coords, int alphabet, void* d_tab16, float* d_rownorm,
                             uint32_t* d_unsafe, hipStream_t s);
hipError_t hs_launch_qprep(const double* d_centers, uint32_t nq, int k, double r2, void* d_c16,
                           uint32_t* d_unsafe, hipStream_t s);
// items[j] for joined segments (>= min_q probing queries and >= min_m members), 0 otherwise, and
// nslices[ql] = 0 for the probes of joined segments; stats[0] += MFMA pairs issued, [1] += real pairs
hipError_t hs_launch_seg_route(const uint64_t* d_seg_key, const uint32_t* d_seg_cnt,
                               const uint32_t* d_seg_qoff, const uint32_t* d_n_seg,
                               const uint32_t* d_sorted_ql, const uint32_t* d_qcount, uint32_t n_max,
                               uint32_t min_q, uint32_t min_m, uint32_t jm, int L, int shift,
                               uint32_t max_q_resident /* segments up to this many queries are issued in
                               16-query column tiles (hs_join8r_kernel); 0: none */,
                               uint32_t* d_items, unsigned long long* d_stats, uint32_t* d_nslices,
                               const uint32_t* d_seg_of, hipStream_t s);
hipError_t hs_launch_item_desc(const hs_tables_dev& tabs, const uint64_t* d_seg_key,
                               const uint32_t* d_seg_cnt,
                               const uint32_t* d_seg_qoff, const uint32_t* d_item_off, uint32_t n_max,
                               const uint32_t* d_sorted_ql, const uint32_t* d_qcount, uint32_t n_items,
                               uint32_t jm, int shift, const uint32_t* d_order, int PW,
                               const uint32_t* d_n_items /* null, or the device-side count: n_items is
                               then the capacity */, uint4* d_desc, hipStream_t s);
// item numbering order of the segments: many-query segments first (stable two-class partition):
// d_big[n + 1] flags (last = 0) -> exclusive scan -> d_order[n], d_items_ordered[n]
// ... and LAST the segments with at most max_q_resident probing queries (d_res flags, null / 0: no such
// class): their items are the tail [split[0], split[1]) of the item list, taken by the query-resident
// join kernel (hs_join8r_kernel)
hipError_t hs_launch_seg_big(const uint32_t* d_seg_cnt, const uint32_t* d_items, uint32_t n,
                             uint32_t min_q, uint32_t max_q_resident, uint32_t* d_big, uint32_t* d_res,
                             hipStream_t s);
hipError_t hs_launch_seg_order(const uint32_t* d_big_pos, const uint32_t* d_res_pos, const uint32_t* d_items,
                               uint32_t n, uint32_t* d_order, uint32_t* d_items_ordered, hipStream_t s);
hipError_t hs_launch_item_split(const uint32_t* d_item_off, const uint32_t* d_res_pos, uint32_t n,
                                uint32_t* d_split, hipStream_t s);
// queries of a segment the query-resident join keeps in registers (two 32-query tiles: with three the
// kernel spills at two waves per SIMD)
#define HS_JR_MAXQ 64u
// ... and alphabets of up to this many residues (its pair table has eight copies of 32 x alphabet entries in LDS)
#define HS_JR_MAX_ALPHABET 24
// members per work item: 512 (one workgroup tile) for the staged kernels, 128 (one wave) for the
// wave-independent int8 join
#define HS_JM_BLOCK 512u
#define HS_JM_WAVE 128u
// queries per work item of the wave-independent join (a wave re-uses its members' operands over
// all of them; the staged kernel's items stop at 2048)
#ifndef HS_JQG_WAVE
#define HS_JQG_WAVE 8192u
#endif
hipError_t hs_launch_gather_c16(const void* d_c16, const uint32_t* d_sorted_ql, uint32_t nql, int L,
                                void* d_out, hipStream_t s);
hipError_t hs_launch_join(const uint4* d_desc, uint32_t n_items, const uint4* d_packed_base,
                          const uint32_t* d_sorted_ql, const void* d_c16s, const void* d_tab16,
                          const float* d_rownorm, int k, uint32_t* d_prov_count, uint32_t prov_cap,
                          uint2* d_prov, int n_blocks, hipStream_t s);
// int8 form of the join filter (hs_join8.hip)
// `wide` below: rows of short k-mers (k <= 20) carry all 8 coordinates on one scale (6 k-steps, table
// d_tabW, scale[4..6]); the launchers that take d_tab8 expect d_tabW in its place then
hipError_t hs_launch_jtables8(const double* d_coords, int alphabet, void* d_tab8, float* d_scale,
                              uint32_t* d_unsafe, void* d_tabR, void* d_tabW, hipStream_t s);
// d_c8b (may be null): the second row per query (columns 4..7 + the refinement's scalars)
// the same rows for queries that are k-mers given as codes (self-join): x^ from the tables, no doubles
hipError_t hs_launch_qprep8_codes(const uint8_t* d_qcodes, uint32_t nq, int k, int wide, double r2,
                                  const double* d_coords, const void* d_tab8, const void* d_tabR,
                                  const void* d_tabW, const float* d_scale, void* d_c8, void* d_c8b,
                                  hipStream_t s);
hipError_t hs_launch_qprep8(const double* d_centers, uint32_t nq, int k, int wide, double r2,
                            const float* d_scale, void* d_c8, uint32_t* d_unsafe, void* d_c8b,
                            hipStream_t s);
// survivors of the 4-column bound -> those that also pass the 8-column bound (compacted, direct form)
hipError_t hs_launch_refine8(const hs_tables_dev& tabs, const uint2* d_prov, const uint32_t* d_prov_count,
                             uint32_t prov_cap, const uint32_t* d_sorted_ql, const void* d_c8,
                             const void* d_c8b, const void* d_tabR, const float* d_scale, int k, int L,
                             const uint32_t* d_qstart, const uint32_t* d_qcount,
                             uint2* d_out, uint32_t* d_out_count, hipStream_t s);
hipError_t hs_launch_gather_c8t(const void* d_c8, const uint32_t* d_sorted_ql, const uint32_t* d_seg_qoff,
                                const uint32_t* d_seg_of, uint32_t nql, int L, int k, int wide, void* d_out,
                                hipStream_t s);
// bytes of a quantised int8 row (32 per k-step: 128 for k <= 25, 192 for k <= 41 and for wide rows,
// 256 for k <= 50) and the members of one work item of the wave-independent int8 join (128 / 64)
int hs_join8_row_bytes(int k, int wide);
uint32_t hs_join8_members_per_item(int k, int wide);
hipError_t hs_launch_join8w(const uint4* d_desc, uint32_t n_items, const uint4* d_packed_base,
                            const uint4* d_rec_base, const void* d_c8t, const void* d_tab8, int k, int wide,
                            uint32_t* d_prov_count, uint32_t prov_cap, uint2* d_prov,
                            uint32_t* d_item_counter, int n_blocks, const uint32_t* d_n_items,
                            double pairs_per_item, uint32_t xcd_run, hipStream_t s, uint32_t chunk = 0);
// the item list's tail [d_split[0], d_split[1]) (segments with <= HS_JR_MAXQ probing queries; k <= 25,
// 4-column rows) through the query-resident form; d_cn_rep = 128 copies of the gamma slots' constant
// factors (HS_J8_CONST bytes); the packed / record arrays must be readable 128 entries past their end
hipError_t hs_launch_join8r(const uint4* d_desc, uint32_t desc_cap, const uint32_t* d_split,
                            const uint4* d_packed_base, const uint32_t* d_rho_base /* the records in four bytes */,
                            const void* d_c8t,
                            const void* d_tab8, int alphabet /* <= HS_JR_MAX_ALPHABET */, uint32_t* d_prov_count, uint32_t prov_cap,
                            uint2* d_prov, uint32_t* d_item_counter, int n_blocks, double pairs_per_item,
                            hipStream_t s);
// bucket-ordered packed copy of one table (k <= 25) + the per-entry 16-byte A-row tails of the
// int8 join (d_out_rec[i] belongs to d_out_packed[i])
hipError_t hs_launch_gather_rec8(const uint4* d_packed_all, const uint32_t* d_ids_sorted, uint32_t n,
                                 int k, int wide, const void* d_tab8, const void* d_tabW, const float* d_scale,
                                 uint4* d_out_packed, uint4* d_out_rec,
                                 uint32_t* d_out_rho /* null, or the record in four bytes (hs_join8r_kernel) */,
                                 hipStream_t s);
hipError_t hs_launch_kth_min(const float* d_slice_min, uint32_t nq, uint32_t per_q, uint32_t topk,
                             float* d_thr, hipStream_t s);
hipError_t hs_launch_topk_exact(const uint8_t* d_codes, const double* d_centers,
                                const double* d_coords, const uint2* d_prov,
                                const uint32_t* d_prov_count, uint32_t prov_cap, int k,
                                uint64_t* d_key, uint64_t* d_val, hipStream_t s);
hipError_t hs_launch_bf_finalize(const uint8_t* d_codes, const double* d_centers,
                                 const double* d_coords, const uint2* d_prov,
                                 const uint32_t* d_prov_count, uint32_t prov_cap, int k, double R,
                                 uint32_t q_base, uint32_t* d_hit_count, uint32_t hit_cap,
                                 uint64_t* d_hit_key, uint64_t* d_hit_val, hipStream_t s);

#endif
