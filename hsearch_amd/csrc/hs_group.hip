// hs_group.hip -- index build, SURVEY 8(a) a7: grouping one table's n k-mers by their HashKey.
//
// The reference inserts every k-mer into an unordered_map keyed by the HashKey string
// (motif_both_points.cpp:212-218, hclust2.cpp:74-84).  The index wants, per table, the ids grouped by
// key -- ascending inside a bucket -- and the distinct keys' fingerprints in ascending order (the
// directory a probe searches).  Round 1 / 2 radix-sorted all n (64-bit fingerprint, id) pairs with
// rocPRIM: 6 - 8 passes of 24 bytes per pair at ~ 0.8 TB/s on this part (ROCm 7.2 ships no gfx950
// tuning for it), two thirds of the build's device time.
//
// Buckets are FEW (10^5 of 10^7 k-mers at configs[1], 4 10^6 of 10^8 at configs[2]'s shape) and very
// skewed (the largest holds 6.5 % of the database), so here the fingerprints are never sorted as such:
//   1. every k-mer's key goes into an open-addressing table of 64-bit fingerprints (one 8-byte slot read per
//      k-mer in the common case, one CAS per DISTINCT key); the k-mer keeps its slot number;
//   2. the distinct keys' fingerprints are compacted out of the table and sorted -- nb of them, a small
//      sort (rocPRIM, on all 64 bits) -- which gives every slot the RANK of its key in the directory;
//   3. the k-mers carry 32-bit ranks now: a stable LSD radix sort of (rank, id) over ceil(log2 nb) bits,
//      8 bits per pass, written here (histogram per 4096-element tile, one scan over digits x tiles,
//      stable scatter through an LDS-staged tile): 2 - 3 passes of 20 bytes per pair;
//   4. bucket boundaries fall out of the sorted ranks;
//   5. every k-mer's bucket ints are compared with its bucket's tuple (hs_group_check_kernel).
// Bucket membership is exactly the reference's string equality: step 5 accepts a k-mer whose ints equal the
// bucket's tuple or whose HashKey string does (aliased tuples share a fingerprint by construction); two
// strings under one fingerprint are reported and the build repeats with the next seed, as with the sorting form.
// Skew does not matter to any step: the hot key's k-mers read one cached slot, and the LSD passes count
// digits of ranks.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "hs_internal.h"

namespace {

constexpr uint64_t FP_EMPTY = ~0ull;
constexpr int RS_BITS = 8, RS_BINS = 1 << RS_BITS;
constexpr uint32_t RS_TILE = 4096;  // elements per block of the radix passes: 256 threads x 16

inline unsigned blocks_for(uint64_t n, unsigned per = 256) { return (unsigned)((n + per - 1) / per); }

// ---- 1. bucket ints -> fingerprint -> slot of an open-addressing table ---------------------------------
// T[C] (C a power of two >= n, all FP_EMPTY at entry), slot from the fingerprint's top bits, linear probing.
// A slot holds the FULL 64-bit fingerprint of its key: a k-mer belongs to the first slot that holds its
// fingerprint, or claims the first empty one (one CAS per DISTINCT key).  ONE random access per k-mer and no
// second one that depends on it -- rounds 2-3 kept (32 bits, id of a representative) in the slot and proved
// the k-mer's membership on the spot against the representative's bucket ints: two DEPENDENT random accesses,
// 12.7 ms per table at 10^8 k-mers.  A slot never changes once written, so a k-mer first looks at it with a
// plain (cacheable) load -- the hot buckets' slots are read millions of times -- and only an empty or foreign
// value sends it to the atomic path.  The membership PROOF (equal fingerprints must mean equal HashKey
// strings) is a pass of its own once every bucket has its tuple: hs_group_check_kernel.
// slot_of[i] = the slot of k-mer i's key.  flag |= 16: table full, or a fingerprint equal to the empty marker
// (the caller groups this table by sorting instead).
__global__ __launch_bounds__(256) void hs_group_insert_kernel(const int32_t* __restrict__ ints, uint64_t n, int K,
                                                              uint32_t seed, uint64_t* __restrict__ T, uint32_t cmask,
                                                              int shift, uint32_t* __restrict__ slot_of,
                                                              uint32_t* __restrict__ flag) {
  __shared__ int32_t s_t[HS_MAX_K * 256];  // the thread's K ints, [j][thread]
  const uint64_t i0 = (uint64_t)blockIdx.x * 256, i = i0 + threadIdx.x;
  int32_t* t = s_t + threadIdx.x;
  if ((K & 3) == 0) {
    // the block's 256 K ints as ONE coalesced stream of 16-byte pieces (a thread reading its own K / 4 pieces
    // touches lines 4 K bytes apart: 64 lines per load instruction of a wave), transposed through LDS
    const uint32_t K4 = (uint32_t)K >> 2;
    const uint64_t n_here = min((uint64_t)256, n - i0);
    const int4* src = reinterpret_cast<const int4*>(ints + i0 * (uint64_t)K);
    for (uint32_t f = threadIdx.x; f < (uint32_t)n_here * K4; f += 256) {
      const int4 v = src[f];
      const uint32_t r = f / K4, j4 = f - r * K4;
      s_t[256 * (4 * j4) + r] = v.x;
      s_t[256 * (4 * j4 + 1) + r] = v.y;
      s_t[256 * (4 * j4 + 2) + r] = v.z;
      s_t[256 * (4 * j4 + 3) + r] = v.w;
    }
    __syncthreads();
    if (i >= n) return;
  } else {
    if (i >= n) return;
    const int32_t* tg = ints + i * (uint64_t)K;
    for (int j = 0; j < K; ++j) t[256 * j] = tg[j];
  }
  uint64_t hk = hs_key_init(seed);
  for (int j = 0; j < K; ++j) hk = hs_key_put_int(hk, t[256 * j]);
  const uint64_t fp = hs_key_fin(hk);
  uint32_t s = (uint32_t)(fp >> shift) & cmask;
  if (fp == FP_EMPTY) {  // (2^-64: the marker itself)
    atomicOr(flag, 16u);
    slot_of[i] = 0;
    return;
  }
  // (bounded: the table is at most as full as distinct keys / k-mers; a chain of thousands means nearly
  // every key is distinct -- a tiny W -- and the build takes the sorting path)
  const uint32_t max_probe = cmask < 4095u ? cmask : 4095u;
  for (uint32_t probe = 0; probe <= max_probe; ++probe) {
    uint64_t w = T[s];  // plain load: a value other than the marker is final
    if (w == FP_EMPTY) {
      w = atomicCAS(reinterpret_cast<unsigned long long*>(&T[s]), (unsigned long long)FP_EMPTY, (unsigned long long)fp);
      if (w == FP_EMPTY) w = fp;  // claimed
    }
    if (w == fp) {
      slot_of[i] = s;
      return;
    }
    s = (s + 1) & cmask;
  }
  atomicOr(flag, 16u);  // table (nearly) full: the caller falls back
  slot_of[i] = 0;
}

// The exact-membership proof, in id order: k-mer i was put into the bucket of rank rank_of[i] because its
// fingerprint is that bucket's; its HashKey string must be the bucket's too (the tuple of the bucket's first
// member).  Identical ints in the common case (the tuple rows of the hot buckets stay in cache, the k-mer's
// own ints are read in order); ints that differ are compared as strings -- aliased tuples ((1,23) and (12,3))
// share string and fingerprint by construction -- and two strings under one fingerprint raise flag |= 1: the
// caller rebuilds with the next seed.
__global__ __launch_bounds__(256) void hs_group_check_kernel(const int32_t* __restrict__ ints, uint64_t n, int K,
                                                             const uint32_t* __restrict__ rank_of,
                                                             const int32_t* __restrict__ dir_tuple,
                                                             uint32_t* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t* px = ints + i * (uint64_t)K;
  const int32_t* py = dir_tuple + (uint64_t)rank_of[i] * K;
  bool same = true;
  for (int j = 0; j < K; ++j) same = same && px[j] == py[j];
  if (same) return;
  int32_t x[HS_MAX_K], y[HS_MAX_K];
  for (int j = 0; j < K; ++j) {
    x[j] = px[j];
    y[j] = py[j];
  }
  if (!hs_key_equal(x, y, K)) atomicOr(flag, 1u);
}
// The same for K a multiple of 4 (the usual 16, 20): one thread per 16-BYTE PIECE of the bucket ints, so that
// the n K ints are read as one coalesced stream (one thread per k-mer reads K / 4 pieces at a stride of 4 K
// bytes: every load instruction of a wave touched 64 different lines, 6.4 ms per table at 10^8 k-mers).  A
// piece that differs from the tuple's sends its thread to the comparison of the whole strings.
__global__ __launch_bounds__(256) void hs_group_check4_kernel(const int4* __restrict__ ints4, uint64_t n_pieces, int K4,
                                                              const uint32_t* __restrict__ rank_of,
                                                              const int4* __restrict__ tuple4,
                                                              uint32_t* __restrict__ flag) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_pieces) return;
  const uint64_t i = t / (uint32_t)K4;
  const uint32_t j4 = (uint32_t)(t - i * (uint32_t)K4);
  const uint64_t rb = (uint64_t)rank_of[i] * (uint32_t)K4;
  const int4 x = ints4[t], y = tuple4[rb + j4];
  if (x.x == y.x && x.y == y.y && x.z == y.z && x.w == y.w) return;
  int32_t xs[HS_MAX_K], ys[HS_MAX_K];
  const int32_t* px = reinterpret_cast<const int32_t*>(ints4 + i * (uint32_t)K4);
  const int32_t* py = reinterpret_cast<const int32_t*>(tuple4 + rb);
  for (int j = 0; j < 4 * K4; ++j) {
    xs[j] = px[j];
    ys[j] = py[j];
  }
  if (!hs_key_equal(xs, ys, 4 * K4)) atomicOr(flag, 1u);
}

// ---- 2. distinct keys out of the table ----------------------------------------------------------------
__global__ __launch_bounds__(256) void hs_fp_count_kernel(const uint64_t* __restrict__ T, uint32_t C,
                                                          uint32_t* __restrict__ blk_cnt) {
  // one block per 1024 slots
  const uint32_t base = blockIdx.x * 1024u;
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t s = base + threadIdx.x + 256u * j;
    c += (s < C && T[s] != FP_EMPTY) ? 1u : 0u;
  }
  for (int off = 32; off; off >>= 1) c += __shfl_xor(c, off);
  __shared__ uint32_t sw[4];
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
}

// dk = the fingerprint the slot holds, ds = the slot
__global__ __launch_bounds__(256) void hs_fp_compact_kernel(const uint64_t* __restrict__ T, uint32_t C,
                                                            const uint32_t* __restrict__ blk_off,
                                                            uint64_t* __restrict__ dk, uint32_t* __restrict__ ds) {
  const uint32_t base = blockIdx.x * 1024u;
  __shared__ uint32_t s_run;
  if (threadIdx.x == 0) s_run = blk_off[blockIdx.x];
  __syncthreads();
  __shared__ uint32_t sw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // 256 consecutive slots per round, in slot order
    const uint32_t s = base + 256u * j + threadIdx.x;
    const uint64_t v = s < C ? T[s] : FP_EMPTY;
    const bool live = v != FP_EMPTY;
    const unsigned long long m = __ballot(live);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sw[w] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0;
    for (int x = 0; x < w; ++x) before += sw[x];
    const uint32_t total = sw[0] + sw[1] + sw[2] + sw[3];
    if (live) {
      const uint32_t o = s_run + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      dk[o] = v;
      ds[o] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_run += total;
    __syncthreads();
  }
}

// rank_of_slot[ds_sorted[r]] = r
__global__ __launch_bounds__(256) void hs_rank_slots_kernel(const uint32_t* __restrict__ ds_sorted, uint32_t nb,
                                                            uint32_t* __restrict__ rank_of_slot) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r < nb) rank_of_slot[ds_sorted[r]] = r;
}

// slot_of[i] <- rank of k-mer i's key (in place)
__global__ __launch_bounds__(256) void hs_rank_kmers_kernel(uint32_t* __restrict__ slot_of, uint64_t n,
                                                            const uint32_t* __restrict__ rank_of_slot) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) slot_of[i] = rank_of_slot[slot_of[i]];
}

// ---- 3. stable LSD radix pass over (rank, id): 8 bits ------------------------------------------------
// A block owns a CONTIGUOUS range of tiles (tiles_per_block of them), so the digit x block histogram that
// has to be scanned is 256 x RS_BLOCKS words whatever n is (one scan over digits x tiles -- 6 10^6 words at
// n = 10^8 -- took rocPRIM's look-back scan longer than the scatter itself), and the scatter walks its tiles
// in order with the running start of every digit in LDS.
// hist[d * n_blocks + b] = elements of block b's tiles with digit d
__global__ __launch_bounds__(256) void hs_rs_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, int shift,
                                                         uint32_t tiles_per_block, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[RS_BINS];
  h[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t tl = 0; tl < tiles_per_block; ++tl) {
    const uint64_t base = ((uint64_t)blockIdx.x * tiles_per_block + tl) * RS_TILE;
    if (base >= n) break;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint64_t e = base + 256u * j + threadIdx.x;
      if (e < n) atomicAdd(&h[(keys[e] >> shift) & (RS_BINS - 1)], 1u);
    }
  }
  __syncthreads();
  hist[(uint64_t)threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}

// Element order inside a tile (= the order the sort must keep): wave w holds elements w * 1024 ..
// w * 1024 + 1023 in 16 chunks of 64 consecutive elements.  Rank of an element among the tile's
// elements with its digit = (elements of earlier waves) + (of earlier chunks of its wave) + (of
// lower lanes of its chunk: a ballot per digit bit).
__global__ __launch_bounds__(256) void hs_rs_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                            const uint32_t* __restrict__ ids_in /* null: iota */,
                                                            uint32_t n, int shift, uint32_t tiles_per_block,
                                                            const uint32_t* __restrict__ hist_scanned,
                                                            uint32_t* __restrict__ keys_out,
                                                            uint32_t* __restrict__ ids_out) {
  __shared__ uint32_t wh[4][RS_BINS];   // per wave: running count per digit, then the wave's exclusive prefix
  __shared__ uint32_t tstart[RS_BINS];  // tile-local start of every digit's run
  __shared__ uint32_t gbase[RS_BINS];   // global start of the tile's run of every digit
  __shared__ uint2 stage[RS_TILE];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  gbase[tid] = hist_scanned[(uint64_t)tid * gridDim.x + blockIdx.x];
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (uint32_t tl = 0; tl < tiles_per_block; ++tl) {
    const uint64_t tile_base = ((uint64_t)blockIdx.x * tiles_per_block + tl) * RS_TILE;
    if (tile_base >= n) break;  // (block-uniform)
    for (int d = tid; d < 4 * RS_BINS; d += 256) (&wh[0][0])[d] = 0;
    __syncthreads();
    const uint32_t base = (uint32_t)tile_base + (uint32_t)w * 1024u;
    uint32_t key[16], id[16], loc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const uint32_t e = base + 64u * c + (uint32_t)lane;
      const bool live = e < n;
      key[c] = live ? keys_in[e] : 0xffffffffu;
      id[c] = live ? (ids_in ? ids_in[e] : e) : 0xffffffffu;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      // (elements past the end count as digit 255: they sit at the very end of the last tile, so every
      // live element of that digit ranks before them, and they are not written)
      const uint32_t e = base + 64u * c + (uint32_t)lane;
      const uint32_t d = e < n ? (key[c] >> shift) & (RS_BINS - 1) : (uint32_t)(RS_BINS - 1);
      unsigned long long m = ~0ull;
#pragma unroll
      for (int b = 0; b < RS_BITS; ++b) {
        const unsigned long long bal = __ballot((d >> b) & 1u);
        m &= ((d >> b) & 1u) ? bal : ~bal;
      }
      const uint32_t before = wh[w][d];  // (every lane of the digit reads it before its leader adds)
      const uint32_t rank = (uint32_t)__popcll(m & lt);
      __builtin_amdgcn_wave_barrier();
      if (rank == 0) wh[w][d] = before + (uint32_t)__popcll(m);
      __builtin_amdgcn_wave_barrier();
      loc[c] = before + rank;
    }
    __syncthreads();
    uint32_t tile_cnt;
    {  // per digit: the waves' counts -> exclusive prefix over the waves; tile total
      const uint32_t c0 = wh[0][tid], c1 = wh[1][tid], c2 = wh[2][tid], c3 = wh[3][tid];
      wh[0][tid] = 0;
      wh[1][tid] = c0;
      wh[2][tid] = c0 + c1;
      wh[3][tid] = c0 + c1 + c2;
      tile_cnt = c0 + c1 + c2 + c3;
      tstart[tid] = tile_cnt;
    }
    __syncthreads();
    if (w == 0) {  // exclusive scan of the 256 tile totals by one wave: 4 digits per lane
      uint32_t v[4], sum = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = tstart[4 * lane + j];
        sum += v[j];
      }
      uint32_t inc = sum;
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
      }
      uint32_t run = inc - sum;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        tstart[4 * lane + j] = run;
        run += v[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const uint32_t e = base + 64u * c + (uint32_t)lane;
      if (e < n) {
        const uint32_t d = (key[c] >> shift) & (RS_BINS - 1);
        stage[tstart[d] + wh[w][d] + loc[c]] = make_uint2(key[c], id[c]);
      }
    }
    __syncthreads();
    const uint32_t live_in_tile = (uint32_t)min((uint64_t)RS_TILE, (uint64_t)n - tile_base);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t p = 256u * j + (uint32_t)tid;
      if (p < live_in_tile) {
        const uint2 v = stage[p];
        const uint32_t d = (v.x >> shift) & (RS_BINS - 1);
        const uint32_t o = gbase[d] + (p - tstart[d]);
        keys_out[o] = v.x;
        ids_out[o] = v.y;
      }
    }
    __syncthreads();
    // (the dead elements of the last tile were counted as digit 255: nothing follows them)
    gbase[tid] += tile_cnt;
  }
}

// ---- 4. boundaries, bucket sizes, the string-equality proof ------------------------------------------
__global__ __launch_bounds__(256) void hs_dir_start_kernel(const uint32_t* __restrict__ ranks_sorted, uint32_t n,
                                                           uint32_t nb, uint32_t* __restrict__ dir_start) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) dir_start[nb] = n;
  if (i >= n) return;
  const uint32_t r = ranks_sorted[i];
  if (i == 0 || ranks_sorted[i - 1] != r) dir_start[r] = i;
}

__global__ __launch_bounds__(256) void hs_dir_max_kernel(const uint32_t* __restrict__ dir_start, uint32_t nb,
                                                         uint32_t* __restrict__ out_max) {
  uint32_t m = 0;
  for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < nb; r += gridDim.x * 256) m = max(m, dir_start[r + 1] - dir_start[r]);
  for (int off = 32; off; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out_max, m);
}

__global__ __launch_bounds__(256) void hs_iota_u32_kernel(uint32_t* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = i;
}

}  // namespace

hipError_t hs_launch_iota_u32(uint32_t* d_out, uint32_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hs_iota_u32_kernel<<<blocks_for(n), 256, 0, s>>>(d_out, n);
  return hipGetLastError();
}

// slots of the fingerprint table for n k-mers: a power of two >= n (load factor = distinct keys / slots
// <= 1 always, a few per cent on the benchmark shapes)
uint32_t hs_group_table_slots(uint64_t n) {
  uint32_t c = 1024;
  while ((uint64_t)c < n && c < (1u << 31)) c <<= 1;
  return c;
}

// blocks of the radix passes: each owns ceil(tiles / blocks) consecutive tiles
constexpr uint32_t RS_BLOCKS = 2048;
uint32_t hs_rs_blocks(uint64_t n) {
  const uint64_t tiles = (n + RS_TILE - 1) / RS_TILE;
  return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(tiles, RS_BLOCKS));
}
static uint32_t rs_tiles_per_block(uint64_t n) {
  const uint64_t tiles = (n + RS_TILE - 1) / RS_TILE;
  const uint32_t nb = hs_rs_blocks(n);
  return (uint32_t)((tiles + nb - 1) / nb);
}

// d_table: C words, the slots
hipError_t hs_launch_group_insert(const int32_t* d_ints, uint64_t n, int K, uint32_t seed, uint64_t* d_table,
                                  uint32_t C, uint32_t* d_slot_of, uint32_t* d_flag, hipStream_t s) {
  hipError_t e = hipMemsetAsync(d_table, 0xff, (size_t)C * 8, s);
  if (e != hipSuccess || !n) return e;
  int log2c = 0;
  while ((1u << log2c) < C) ++log2c;
  hs_group_insert_kernel<<<blocks_for(n), 256, 0, s>>>(d_ints, n, K, seed, d_table, C - 1, 64 - log2c, d_slot_of,
                                                       d_flag);
  return hipGetLastError();
}

hipError_t hs_launch_group_check(const int32_t* d_ints, uint64_t n, int K, const uint32_t* d_rank_of,
                                 const int32_t* d_dir_tuple, uint32_t* d_flag, hipStream_t s) {
  if (!n) return hipSuccess;
  if ((K & 3) == 0)
    hs_group_check4_kernel<<<blocks_for(n * (uint64_t)(K / 4)), 256, 0, s>>>(
        reinterpret_cast<const int4*>(d_ints), n * (uint64_t)(K / 4), K / 4, d_rank_of,
        reinterpret_cast<const int4*>(d_dir_tuple), d_flag);
  else
    hs_group_check_kernel<<<blocks_for(n), 256, 0, s>>>(d_ints, n, K, d_rank_of, d_dir_tuple, d_flag);
  return hipGetLastError();
}

hipError_t hs_launch_fp_count(const uint64_t* d_table, uint32_t C, uint32_t* d_blk_cnt, hipStream_t s) {
  hs_fp_count_kernel<<<(C + 1023) / 1024, 256, 0, s>>>(d_table, C, d_blk_cnt);
  return hipGetLastError();
}

hipError_t hs_launch_fp_compact(const uint64_t* d_table, uint32_t C, const uint32_t* d_blk_off, uint64_t* d_dk,
                                uint32_t* d_ds, hipStream_t s) {
  hs_fp_compact_kernel<<<(C + 1023) / 1024, 256, 0, s>>>(d_table, C, d_blk_off, d_dk, d_ds);
  return hipGetLastError();
}

hipError_t hs_launch_rank_slots(const uint32_t* d_ds_sorted, uint32_t nb, uint32_t* d_rank_of_slot, hipStream_t s) {
  if (!nb) return hipSuccess;
  hs_rank_slots_kernel<<<blocks_for(nb), 256, 0, s>>>(d_ds_sorted, nb, d_rank_of_slot);
  return hipGetLastError();
}

hipError_t hs_launch_rank_kmers(uint32_t* d_slot_of, uint64_t n, const uint32_t* d_rank_of_slot, hipStream_t s) {
  if (!n) return hipSuccess;
  hs_rank_kmers_kernel<<<blocks_for(n), 256, 0, s>>>(d_slot_of, n, d_rank_of_slot);
  return hipGetLastError();
}

hipError_t hs_launch_rs_hist(const uint32_t* d_keys, uint32_t n, int shift, uint32_t* d_hist, hipStream_t s) {
  hs_rs_hist_kernel<<<hs_rs_blocks(n), 256, 0, s>>>(d_keys, n, shift, rs_tiles_per_block(n), d_hist);
  return hipGetLastError();
}

hipError_t hs_launch_rs_scatter(const uint32_t* d_keys_in, const uint32_t* d_ids_in, uint32_t n, int shift,
                                const uint32_t* d_hist_scanned, uint32_t* d_keys_out, uint32_t* d_ids_out,
                                hipStream_t s) {
  hs_rs_scatter_kernel<<<hs_rs_blocks(n), 256, 0, s>>>(d_keys_in, d_ids_in, n, shift, rs_tiles_per_block(n),
                                                       d_hist_scanned, d_keys_out, d_ids_out);
  return hipGetLastError();
}

hipError_t hs_launch_dir_start(const uint32_t* d_ranks_sorted, uint32_t n, uint32_t nb, uint32_t* d_dir_start,
                               uint32_t* d_max, hipStream_t s) {
  hs_dir_start_kernel<<<blocks_for(n), 256, 0, s>>>(d_ranks_sorted, n, nb, d_dir_start);
  hs_dir_max_kernel<<<std::min(1024u, blocks_for(nb)), 256, 0, s>>>(d_dir_start, nb, d_max);
  return hipGetLastError();
}

// ---- index build with the hashing spread over ranks ("Index build") ---------------------
// A rank evaluates the hash functions for its block [lo, lo + cnt) of the k-mers only; the fingerprints
// are all-gathered, every rank groups all of them, and the exact-membership proof is done by the rank that
// HAS a k-mer's bucket ints, against the bucket's tuple -- the tuple of the bucket's first member, which
// the rank owning that k-mer contributes (hs_shard_first_tuples_kernel: zeros elsewhere, the ranks' arrays
// are summed).
namespace {

__global__ __launch_bounds__(256) void hs_shard_first_tuples_kernel(const uint32_t* __restrict__ dir_start,
                                                                    const uint32_t* __restrict__ ids,
                                                                    const int32_t* __restrict__ ints_block,
                                                                    uint32_t lo, uint32_t cnt, uint32_t nb, int K,
                                                                    int32_t* __restrict__ tuples) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (uint64_t)nb * K) return;
  const uint32_t bkt = (uint32_t)(t / K);
  const int j = (int)(t % K);
  const uint32_t first = ids[dir_start[bkt]];
  tuples[t] = (first >= lo && first - lo < cnt) ? ints_block[(uint64_t)(first - lo) * K + j] : 0;
}

// own k-mer i (id lo + i): its bucket = the one whose range holds its sorted position; its ints against
// the bucket's tuple.  flag |= 1: equal fingerprints, different HashKey strings.
__global__ __launch_bounds__(256) void hs_shard_check_kernel(const int32_t* __restrict__ ints_block, uint32_t lo,
                                                             uint32_t cnt, int K, const uint32_t* __restrict__ pos_of,
                                                             const uint32_t* __restrict__ dir_start, uint32_t nb,
                                                             const int32_t* __restrict__ dir_tuple,
                                                             uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cnt) return;
  const uint32_t p = pos_of[lo + i];
  uint32_t a = 0, b = nb;  // largest r with dir_start[r] <= p
  while (b - a > 1) {
    const uint32_t mid = (a + b) >> 1;
    if (dir_start[mid] <= p) a = mid; else b = mid;
  }
  const int32_t* px = ints_block + (uint64_t)i * K;
  const int32_t* py = dir_tuple + (uint64_t)a * K;
  bool same = true;
  for (int j = 0; j < K; ++j) same = same && px[j] == py[j];
  if (same) return;
  int32_t x[HS_MAX_K], y[HS_MAX_K];
  for (int j = 0; j < K; ++j) {
    x[j] = px[j];
    y[j] = py[j];
  }
  if (!hs_key_equal(x, y, K)) atomicOr(flag, 1u);
}

}  // namespace

hipError_t hs_launch_shard_first_tuples(const uint32_t* d_dir_start, const uint32_t* d_ids, const int32_t* d_ints_block,
                                        uint32_t lo, uint32_t cnt, uint32_t nb, int K, int32_t* d_tuples,
                                        hipStream_t s) {
  if (!nb) return hipSuccess;
  hs_shard_first_tuples_kernel<<<blocks_for((uint64_t)nb * K), 256, 0, s>>>(d_dir_start, d_ids, d_ints_block, lo, cnt,
                                                                            nb, K, d_tuples);
  return hipGetLastError();
}

hipError_t hs_launch_shard_check(const int32_t* d_ints_block, uint32_t lo, uint32_t cnt, int K,
                                 const uint32_t* d_pos_of, const uint32_t* d_dir_start, uint32_t nb,
                                 const int32_t* d_dir_tuple, uint32_t* d_flag, hipStream_t s) {
  if (!cnt) return hipSuccess;
  hs_shard_check_kernel<<<blocks_for(cnt), 256, 0, s>>>(d_ints_block, lo, cnt, K, d_pos_of, d_dir_start, nb,
                                                        d_dir_tuple, d_flag);
  return hipGetLastError();
}
