// hs_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the motif-search hot path.
//
// Built with -ffp-contract=off: the reference evaluates its fp64 sums with separate multiply and
// add (lsh.hpp:33-42, motif_both_points.cpp:176-183; built ISO C++11 without -march, so never
// contracted), and the bucket ints / hit sets must match bit for bit.  The exact paths also spell
// the operations as __dmul_rn/__dadd_rn/__dsub_rn/__ddiv_rn/__dsqrt_rn so no flag can fuse them.
//
// Kernels
//   hs_embed_kernel      a2  codes -> R^{8k} doubles                       (HBM write bound)
//   hs_hash_kernel       a4+a5 exact fp64 projections + floor((dot+b)/W)   (fp64 VALU bound)
//   hs_keys_kernel       a6  fingerprint of the HashKey character stream
//   hs_check_runs_kernel a6  exact string-equality check of fingerprint runs (build)
//   hs_pack_kernel           5-bit residue packing, 25 residues per 16-byte word
//   hs_gather_packed_kernel  bucket-ordered packed copies (build)
//   hs_probe_kernel      a8  directory lookup, exact key-string check
//   hs_qtables_kernel    a9  per-query tables T[pos][aa] = |c_pos - coord[aa]|^2 (fp32)
//   hs_verify_kernel     a9  candidate scan: coalesced 16-B loads, tables held ACROSS LANES in
//                            VGPRs, ds_bpermute lookups, wave ballot compaction    (HBM bound)
//   hs_finalize_kernel   a8-a10 first-seen dedupe, exact left-to-right fp64 d2, d2 <= R^2
#include <algorithm>

#include "hs_internal.h"

namespace {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (WAVE - 1)); }

__device__ __forceinline__ double exact_dist2(const uint8_t* __restrict__ row,
                                              const double* __restrict__ c,
                                              const double* __restrict__ coords, int k);

// ------------------------------------------------------------------------------------------ embed
__global__ __launch_bounds__(256) void hs_embed_kernel(const uint8_t* __restrict__ codes,
                                                       uint64_t total, const double* __restrict__ coords,
                                                       double* __restrict__ out) {
  uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (; idx < total; idx += stride) out[idx] = coords[(int)codes[idx >> 3] * 8 + (int)(idx & 7)];
}

// ------------------------------------------------------------------------------------------- hash
// One thread = one point; KC accumulators = KC hash functions at a time.  The plane values are
// wave-uniform (scalar loads) and come from the TRANSPOSED plane matrix aT[i][f] (dimension-major:
// the KC values of one dimension are one contiguous scalar load), so consecutive vector
// instructions belong to KC different accumulators -- KC independent dependency chains -- and the
// next dimension's planes are fetched while the current one is consumed.  The point's coordinates
// come from the 20x8 table in LDS (codes) or from its own row (arbitrary points).  Per function
// the order is still strictly i = 0..d-1, product rounded, then sum rounded -- lsh.hpp:33-42 --
// then floor((dot + b) / W) -- lsh.hpp:44-49.
template <int KC, bool FROM_CODES>
__global__ __launch_bounds__(256) void hs_hash_kernel(const uint8_t* __restrict__ codes,
                                                      const double* __restrict__ pts, uint64_t n,
                                                      int k, const double* __restrict__ aT, int ldf,
                                                      const double* __restrict__ b, int F, double W,
                                                      const double* __restrict__ coords,
                                                      int32_t* __restrict__ out, int out_stride) {
  __shared__ double s_coords[HS_ALPHABET_PAD * 8];
  extern __shared__ uint8_t s_codes[];  // [256][k]
  const int d = 8 * k;
  const uint64_t base = (uint64_t)blockIdx.x * 256;
  const uint64_t i = base + threadIdx.x;
  const bool valid = i < n;
  if (FROM_CODES) {
    for (int t = threadIdx.x; t < HS_ALPHABET_PAD * 8; t += 256) s_coords[t] = coords[t];
    const uint64_t lo = base * k;
    const uint64_t hi = (base + 256 < n ? base + 256 : n) * (uint64_t)k;
    for (uint64_t t = lo + threadIdx.x; t < lo + 256ull * k; t += 256)
      s_codes[t - lo] = t < hi ? codes[t] : (uint8_t)0;
    __syncthreads();
  }
  const double* xrow = FROM_CODES ? nullptr : pts + (valid ? i : 0) * (uint64_t)d;
  // gridDim.y > 1 (few points, e.g. a query batch): each block row takes every gridDim.y-th chunk
  // of KC functions, so the launch fills the chip instead of running one block per CU.
  for (int fc = (int)blockIdx.y * KC; fc < F; fc += KC * (int)gridDim.y) {
    double acc[KC];
#pragma unroll
    for (int f = 0; f < KC; ++f) acc[f] = 0.0;
    const double* ar = aT + fc;  // row i of this chunk: ar + i * ldf (wave-uniform)
    double p0[KC], p1[KC];       // planes of two consecutive dimensions (scalar registers)
#pragma unroll
    for (int f = 0; f < KC; ++f) p0[f] = ar[f];
    double xn[8];  // arbitrary points: the next position's coordinates, fetched one position ahead
    if (!FROM_CODES) {
#pragma unroll
      for (int j = 0; j < 8; ++j) xn[j] = xrow[j];
    }
    for (int pos = 0; pos < k; ++pos) {
      double x[8];
      if (FROM_CODES) {
        const int c = s_codes[threadIdx.x * k + pos] & (HS_ALPHABET_PAD - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = s_coords[c * 8 + j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = xn[j];
        const int pn = min(pos + 1, k - 1);  // past the end: a harmless re-read
#pragma unroll
        for (int j = 0; j < 8; ++j) xn[j] = xrow[8 * pn + j];
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int i = 8 * pos + 2 * jj;
        const double* r1 = ar + (size_t)(i + 1) * ldf;
#pragma unroll
        for (int f = 0; f < KC; ++f) p1[f] = r1[f];
        {  // products first, then sums: no instruction waits for the one before it
          double m[KC];
#pragma unroll
          for (int f = 0; f < KC; ++f) m[f] = __dmul_rn(x[2 * jj], p0[f]);
#pragma unroll
          for (int f = 0; f < KC; ++f) acc[f] = __dadd_rn(acc[f], m[f]);
        }
        const double* r2 = ar + (size_t)min(i + 2, d - 1) * ldf;  // past the end: a harmless re-read
#pragma unroll
        for (int f = 0; f < KC; ++f) p0[f] = r2[f];
        {
          double m[KC];
#pragma unroll
          for (int f = 0; f < KC; ++f) m[f] = __dmul_rn(x[2 * jj + 1], p1[f]);
#pragma unroll
          for (int f = 0; f < KC; ++f) acc[f] = __dadd_rn(acc[f], m[f]);
        }
      }
    }
    if (valid) {
#pragma unroll
      for (int f = 0; f < KC; ++f) {
        const double val = __dadd_rn(acc[f], b[fc + f]);
        out[i * (uint64_t)out_stride + fc + f] = (int32_t)floor(__ddiv_rn(val, W));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------- keys
__global__ __launch_bounds__(256) void hs_keys_kernel(const int32_t* __restrict__ ints, uint64_t n,
                                                      int stride, int K, uint32_t seed,
                                                      uint64_t* __restrict__ keys,
                                                      uint32_t* __restrict__ ids) {
  // the K ints of a point arrive with 16-byte loads, all in flight at once, and wait in LDS
  // (s_t[j][thread]) for the serial character hash
  __shared__ int32_t s_t[HS_MAX_K * 256];
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t* tg = ints + i * (uint64_t)stride;
  int32_t* t = s_t + threadIdx.x;
  if ((K & 3) == 0 && (stride & 3) == 0) {
    int4 v[HS_MAX_K / 4];
#pragma unroll
    for (int j4 = 0; j4 < HS_MAX_K / 4; ++j4)
      if (4 * j4 < K) v[j4] = reinterpret_cast<const int4*>(tg)[j4];
#pragma unroll
    for (int j4 = 0; j4 < HS_MAX_K / 4; ++j4)
      if (4 * j4 < K) {
        t[256 * (4 * j4)] = v[j4].x;
        t[256 * (4 * j4 + 1)] = v[j4].y;
        t[256 * (4 * j4 + 2)] = v[j4].z;
        t[256 * (4 * j4 + 3)] = v[j4].w;
      }
  } else {
    for (int j = 0; j < K; ++j) t[256 * j] = tg[j];
  }
  uint64_t hk = hs_key_init(seed);
  for (int j = 0; j < K; ++j) hk = hs_key_put_int(hk, t[256 * j]);
  keys[i] = hs_key_fin(hk);
  if (ids) ids[i] = (uint32_t)i;
}

// Sorted neighbours with equal fingerprints must have equal key STRINGS.  Identical tuples (the
// common case) are settled here without any scratch; neighbours whose tuples differ (aliased
// strings or a fingerprint collision) are queued for hs_check_runs_slow_kernel.
// slow[0] = queued count, slow[1..] = positions p.
__global__ __launch_bounds__(256) void hs_check_runs_kernel(const uint64_t* __restrict__ keys,
                                                            const uint32_t* __restrict__ ids,
                                                            const int32_t* __restrict__ ints,
                                                            uint64_t n, int K,
                                                            uint32_t* __restrict__ slow,
                                                            uint32_t slow_cap, int sorted_from_bit,
                                                            uint32_t* __restrict__ flag) {
  const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x + 1;
  if (p >= n) return;
  if (keys[p] != keys[p - 1]) {
    // the sort looked at bits [sorted_from_bit, 64) only: two fingerprints that agree there may have
    // interleaved -- flag 4, the caller sorts this table again on all 64 bits
    if (sorted_from_bit && (keys[p] >> sorted_from_bit) == (keys[p - 1] >> sorted_from_bit)) atomicOr(flag, 4u);
    return;
  }
  const int32_t* px = ints + (uint64_t)ids[p] * K;
  const int32_t* py = ints + (uint64_t)ids[p - 1] * K;
  bool same = true;
  if ((K & 3) == 0) {  // 16-byte loads, all in flight at once
    int4 vx[HS_MAX_K / 4], vy[HS_MAX_K / 4];
#pragma unroll
    for (int j4 = 0; j4 < HS_MAX_K / 4; ++j4)
      if (4 * j4 < K) {
        vx[j4] = reinterpret_cast<const int4*>(px)[j4];
        vy[j4] = reinterpret_cast<const int4*>(py)[j4];
      }
#pragma unroll
    for (int j4 = 0; j4 < HS_MAX_K / 4; ++j4)
      if (4 * j4 < K)
        same = same && vx[j4].x == vy[j4].x && vx[j4].y == vy[j4].y && vx[j4].z == vy[j4].z &&
               vx[j4].w == vy[j4].w;
  } else {
    for (int j = 0; j < K; ++j) same = same && (px[j] == py[j]);
  }
  if (same) return;
  const uint32_t at = atomicAdd(slow, 1u);
  if (at < slow_cap) slow[1 + at] = (uint32_t)p;
}

// flag |= 1: a fingerprint collision (equal fingerprints, different key strings);
// flag |= 2: more queued pairs than the queue holds (the caller then checks every pair).
__global__ __launch_bounds__(256) void hs_check_runs_slow_kernel(const uint32_t* __restrict__ ids,
                                                                 const int32_t* __restrict__ ints,
                                                                 int K, const uint32_t* __restrict__ slow,
                                                                 uint32_t slow_cap, uint64_t n_all,
                                                                 const uint64_t* __restrict__ keys,
                                                                 uint32_t* __restrict__ flag) {
  // n_all != 0: exhaustive mode over every neighbour pair (queue overflow)
  const uint64_t total = n_all ? n_all - 1 : min(slow[0], slow_cap);
  if (!n_all && slow[0] > slow_cap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(flag, 2u);
    return;
  }
  for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (uint64_t)gridDim.x * 256) {
    const uint64_t p = n_all ? t + 1 : slow[1 + t];
    if (n_all && keys[p] != keys[p - 1]) continue;
    int32_t x[HS_MAX_K], y[HS_MAX_K];
    const int32_t* px = ints + (uint64_t)ids[p] * K;
    const int32_t* py = ints + (uint64_t)ids[p - 1] * K;
    for (int j = 0; j < K; ++j) {
      x[j] = px[j];
      y[j] = py[j];
    }
    if (!hs_key_equal(x, y, K)) atomicOr(flag, 1u);
  }
}

__global__ __launch_bounds__(256) void hs_dir_tuples_kernel(const uint32_t* __restrict__ dir_start,
                                                            const uint32_t* __restrict__ ids,
                                                            const int32_t* __restrict__ ints,
                                                            uint32_t nb, int K,
                                                            int32_t* __restrict__ dir_tuple) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (uint64_t)nb * K) return;
  const uint32_t bkt = (uint32_t)(t / K);
  const int j = (int)(t % K);
  dir_tuple[t] = ints[(uint64_t)ids[dir_start[bkt]] * K + j];
}

// counts[nb] -> starts[nb+1] is done by the scan primitive; this fills the sentinel.
__global__ void hs_set_u32_kernel(uint32_t* p, uint32_t v) { *p = v; }

__global__ __launch_bounds__(256) void hs_max_u32_kernel(const uint32_t* __restrict__ in, uint32_t n,
                                                         uint32_t* __restrict__ out) {
  uint32_t m = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) m = max(m, in[i]);
  for (int off = 32; off; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
  if (lane_id() == 0) atomicMax(out, m);
}

// ------------------------------------------------------------------------------------------- pack
// 25 residues x 5 bits in each 16-byte word (3 pad bits); a k-mer takes ceil(k/25) words.
__global__ __launch_bounds__(256) void hs_pack_kernel(const uint8_t* __restrict__ codes, uint64_t n,
                                                      int k, int PW, uint32_t alphabet,
                                                      uint4* __restrict__ packed,
                                                      uint32_t* __restrict__ bad) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint8_t* row = codes + i * (uint64_t)k;
  uint32_t any_bad = 0;
  for (int w = 0; w < PW; ++w) {
    uint32_t v[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 25; ++r) {
      const int p = 25 * w + r;
      uint32_t c = p < k ? (uint32_t)row[p] : 0u;
      any_bad |= (c >= alphabet);
      c &= 31u;
      const int bit = 5 * r, wi = bit >> 5, sh = bit & 31;
      v[wi] |= c << sh;
      if (sh > 27) v[wi + 1] |= c >> (32 - sh);
    }
    packed[i * (uint64_t)PW + w] = make_uint4(v[0], v[1], v[2], v[3]);
  }
  if (any_bad) atomicOr(bad, 1u);
}

// hs_query_codes: the caller's query codes into the handle's own buffer, checked on the way -- a byte
// that is no row of the coordinate table raises the flag (the call then fails) and is stored as 0, so
// that no kernel of the batch indexes a table with it.  16 bytes per thread.
__global__ __launch_bounds__(256) void hs_check_codes_kernel(const uint8_t* __restrict__ in, uint64_t n_bytes,
                                                             uint32_t alphabet, uint8_t* __restrict__ out,
                                                             uint32_t* __restrict__ bad) {
  const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (i0 >= n_bytes) return;
  bool any_bad = false;
  const uint64_t i1 = i0 + 16 < n_bytes ? i0 + 16 : n_bytes;
  for (uint64_t i = i0; i < i1; ++i) {
    uint8_t c = in[i];
    if (c >= alphabet) {
      any_bad = true;
      c = 0;
    }
    out[i] = c;
  }
  if (any_bad) atomicOr(bad, 1u);
}

__global__ __launch_bounds__(256) void hs_gather_packed_kernel(const uint4* __restrict__ packed_all,
                                                               const uint32_t* __restrict__ ids,
                                                               uint64_t n, int PW,
                                                               uint4* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * (uint64_t)PW) return;
  const uint64_t p = t / PW;
  const int w = (int)(t % PW);
  out[t] = packed_all[(uint64_t)ids[p] * PW + w];
}

// 4 * residue r of a packed word (the ds_bpermute byte address of lane `residue`)
template <int R>
__device__ __forceinline__ int residue_x4(const uint4& p) {
  constexpr int bit = 5 * R, wi = bit >> 5, sh = bit & 31;
  const uint32_t lo = wi == 0 ? p.x : wi == 1 ? p.y : wi == 2 ? p.z : p.w;
  if constexpr (sh > 27) {
    const uint32_t hi = wi == 0 ? p.y : wi == 1 ? p.z : p.w;
    return (int)((__funnelshift_r(lo, hi, sh) & 31u) << 2);
  } else if constexpr (sh >= 2) {
    return (int)((lo >> (sh - 2)) & 0x7cu);
  } else {
    return (int)((lo << (2 - sh)) & 0x7cu);
  }
}

// ------------------------------------------------------------------------------------------ probe
// Fast path: fingerprint found and the bucket's stored tuple IDENTICAL to the query's (no scratch).
// A found fingerprint with a different tuple (aliased key strings, or a fingerprint collision
// between a query and a DB key) is queued: slow[0] = count, slow[1..] = ql.
__global__ __launch_bounds__(256) void hs_probe_kernel(hs_tables_dev tabs,
                                                       const int32_t* __restrict__ qints,
                                                       uint32_t nq, int K, int L, uint32_t seed,
                                                       uint32_t* __restrict__ qstart,
                                                       uint32_t* __restrict__ qcount,
                                                       uint32_t* __restrict__ nslices,
                                                       uint64_t* __restrict__ cand_out,
                                                       unsigned long long* __restrict__ cand_total,
                                                       uint32_t* __restrict__ slow,
                                                       const uint32_t* __restrict__ dir_base,
                                                       uint32_t nb_total,
                                                       uint32_t* __restrict__ bucket_count,
                                                       uint32_t* __restrict__ qbucket,
                                                       uint32_t* __restrict__ qrank) {
  // The probe is a chain of dependent memory round trips (tuple, 17 directory steps, bucket tuple):
  // the K bucket ints of a probe are fetched with 16-byte loads, all in flight at once, and parked
  // in LDS (s_t[j][thread]: conflict-free), instead of K loads one after the other.
  __shared__ int32_t s_t[HS_MAX_K * 256];
  // (bucket partition with the part's probes listed ahead of the kernel: one thread per list entry)
  const uint32_t slot_ = blockIdx.x * 256 + threadIdx.x;
  const uint32_t ql = tabs.probe_list ? (slot_ < tabs.n_list ? tabs.probe_list[slot_] : 0xffffffffu) : slot_;
  uint32_t count = 0, start = 0;
  // grouping of the probes by bucket (for the bucket join): global bucket number and arrival rank;
  // nb_total = the pseudo-bucket of probes that found none
  uint32_t gb = nb_total;
  bool ranked = false;  // false: out of range, or left to hs_probe_slow_kernel
  if (ql < nq * (uint32_t)L) {
    ranked = true;
    const int l = (int)(ql % (uint32_t)L);
    const int32_t* tg = qints + (uint64_t)ql * K;
    int32_t* t = s_t + threadIdx.x;  // t[256 j] = bucket int j of this probe
    if ((K & 3) == 0) {
      int4 v[HS_MAX_K / 4];
#pragma unroll
      for (int j4 = 0; j4 < HS_MAX_K / 4; ++j4)
        if (4 * j4 < K) v[j4] = reinterpret_cast<const int4*>(tg)[j4];
#pragma unroll
      for (int j4 = 0; j4 < HS_MAX_K / 4; ++j4)
        if (4 * j4 < K) {
          t[256 * (4 * j4)] = v[j4].x;
          t[256 * (4 * j4 + 1)] = v[j4].y;
          t[256 * (4 * j4 + 2)] = v[j4].z;
          t[256 * (4 * j4 + 3)] = v[j4].w;
        }
    } else {
      for (int j = 0; j < K; ++j) t[256 * j] = tg[j];
    }
    const hs_table_dev& tb = tabs.t[l];
    // (bucket partition: a probe of another part is not looked for -- as if the table did not have the bucket;
    // with a probe list the part's own probes were picked ahead of this kernel, by the same rule)
    bool mine = true;
    if (tabs.n_parts > 1u && !tabs.probe_list) {
      const uint64_t th = hs_tuple_hash(t, K, 256);
      uint32_t glo = 0, ghi = tb.n_giant;
      while (glo < ghi) {
        const uint32_t mid = (glo + ghi) >> 1;
        if (tb.giant_key[mid] < th) glo = mid + 1; else ghi = mid;
      }
      const bool giant = glo < tb.n_giant && tb.giant_key[glo] == th;
      mine = hs_probe_part(th, giant, tabs.q_first + ql / (uint32_t)L, tabs.n_parts) == tabs.part;
    }
    uint64_t hk = hs_key_init(seed);
    for (int j = 0; j < K; ++j) hk = hs_key_put_int(hk, t[256 * j]);
    const uint64_t key = hs_key_fin(hk);
    // fingerprints are uniform: their top J bits (2^J >= nb) index a jump table that leaves about
    // one directory entry to look at, instead of log2(nb) dependent round trips
    const uint32_t slot = (uint32_t)(key >> tb.jump_shift);
    uint32_t lo = 0, hi = 0;
    if (mine) {
      lo = tb.dir_jump[slot];
      hi = tb.dir_jump[slot + 1];
    }
    if (tb.dir_rec) {
      // one line per candidate bucket of the slot (about one): all four pieces of the first are asked for
      // together; the fingerprints ascend, so the walk stops at the first one that is not smaller
      bool found = false;
      uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
      for (; lo < hi; ++lo) {
        const uint4* r = tb.dir_rec + 4 * (uint64_t)lo;
        r0 = r[0];
        r1 = r[1];
        r2 = r[2];
        r3 = r[3];
        const uint64_t kr = ((uint64_t)r0.y << 32) | r0.x;
        if (kr >= key) {
          found = kr == key;
          break;
        }
      }
      if (found) {
        const uint32_t w[12] = {r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
        bool same = true;
#pragma unroll
        for (int j = 0; j < HS_REC_MAX_K; ++j)
          if (j < K) same = same && t[256 * j] == (int32_t)(int16_t)(w[j >> 1] >> (16 * (j & 1)));
        if (same) {
          start = r0.z;
          count = r0.w;
          if (dir_base) gb = dir_base[l] + lo;
        } else {
          slow[1 + atomicAdd(slow, 1u)] = ql;
          ranked = false;
        }
      }
    } else {
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (tb.dir_key[mid] < key) lo = mid + 1; else hi = mid;
    }
    if (mine && lo < tb.nb && tb.dir_key[lo] == key) {
      const int32_t* u = tb.dir_tuple + (uint64_t)lo * K;
      bool same = true;
      if ((K & 3) == 0) {
        int4 w[HS_MAX_K / 4];
#pragma unroll
        for (int j4 = 0; j4 < HS_MAX_K / 4; ++j4)
          if (4 * j4 < K) w[j4] = reinterpret_cast<const int4*>(u)[j4];
#pragma unroll
        for (int j4 = 0; j4 < HS_MAX_K / 4; ++j4)
          if (4 * j4 < K)
            same = same && t[256 * (4 * j4)] == w[j4].x && t[256 * (4 * j4 + 1)] == w[j4].y &&
                   t[256 * (4 * j4 + 2)] == w[j4].z && t[256 * (4 * j4 + 3)] == w[j4].w;
      } else {
        for (int j = 0; j < K; ++j) same = same && (t[256 * j] == u[j]);
      }
      if (same) {
        start = tb.dir_start[lo];
        count = tb.dir_start[lo + 1] - start;
        if (dir_base) gb = dir_base[l] + lo;
      } else {
        slow[1 + atomicAdd(slow, 1u)] = ql;
        ranked = false;
      }
    }
    }  // (no directory records)
    qstart[ql] = start;
    qcount[ql] = count;
    nslices[ql] = (count + HS_SLICE - 1) / HS_SLICE;
    if (cand_out) cand_out[ql] = count;
  }
  // (bucket_count == null with qbucket set: the caller groups the probes by sorting them on their
  // bucket number -- hs_launch_seg_group_sparse -- and needs no ranks)
  if (!bucket_count && qbucket && ranked) qbucket[ql] = gb;
  if (bucket_count) {
    const bool pseudo = ranked && gb == nb_total;
    // one counter access per BLOCK for the pseudo-bucket (same-address atomics are slow): ranks
    // inside the block from an LDS counter, the block's base from the global one
    __shared__ uint32_t s_pseudo, s_pbase;
    if (threadIdx.x == 0) s_pseudo = 0;
    __syncthreads();
    uint32_t rank = 0;
    if (pseudo) rank = atomicAdd(&s_pseudo, 1u);
    __syncthreads();
    if (threadIdx.x == 0 && s_pseudo) s_pbase = atomicAdd(&bucket_count[nb_total], s_pseudo);
    __syncthreads();
    if (pseudo) rank += s_pbase;
    if (ranked && !pseudo) rank = atomicAdd(&bucket_count[gb], 1u);
    if (ranked) {
      qbucket[ql] = gb;
      qrank[ql] = rank;
    }
  }
  // candidate total: one access to the global counter per BLOCK (same-address atomics deliver
  // ~100 per microsecond chip-wide: one per wave -- 12 500 of them -- was most of this kernel)
  unsigned long long c = count;
  for (int off = 32; off; off >>= 1) c += __shfl_xor(c, off);
  __shared__ unsigned long long s_c[4];
  if (lane_id() == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = s_c[0] + s_c[1] + s_c[2] + s_c[3];
    if (t) atomicAdd(cand_total, t);
  }
}

// Self-join: query q IS the indexed k-mer first_id + q, so the bucket it probes in table l is the
// bucket it sits in -- found from its sorted position (pos_of) by a binary search over the bucket
// boundaries, no hashing and no directory lookup.  Same outputs as hs_probe_kernel.
__global__ __launch_bounds__(256) void hs_self_probe_kernel(hs_tables_dev tabs, uint32_t first_id, uint32_t nq,
                                                            int L, uint32_t* __restrict__ qstart,
                                                            uint32_t* __restrict__ qcount,
                                                            uint32_t* __restrict__ nslices,
                                                            uint64_t* __restrict__ cand_out,
                                                            unsigned long long* __restrict__ cand_total,
                                                            const uint32_t* __restrict__ dir_base,
                                                            uint32_t* __restrict__ bucket_count,
                                                            uint32_t* __restrict__ qbucket,
                                                            uint32_t* __restrict__ qrank) {
  const uint32_t ql = blockIdx.x * 256 + threadIdx.x;
  uint32_t count = 0;
  if (ql < nq * (uint32_t)L) {
    const uint32_t q = ql / (uint32_t)L;
    const int l = (int)(ql % (uint32_t)L);
    const hs_table_dev& tb = tabs.t[l];
    const uint32_t p = tb.pos_of[first_id + q];
    uint32_t lo = 0, hi = tb.nb;  // largest b with dir_start[b] <= p
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (tb.dir_start[mid] <= p) lo = mid; else hi = mid;
    }
    const uint32_t start = tb.dir_start[lo];
    count = tb.dir_start[lo + 1] - start;
    qstart[ql] = start;
    qcount[ql] = count;
    nslices[ql] = (count + HS_SLICE - 1) / HS_SLICE;
    if (cand_out) cand_out[ql] = count;
    if (bucket_count) {
      const uint32_t gb = dir_base[l] + lo;
      qbucket[ql] = gb;
      qrank[ql] = atomicAdd(&bucket_count[gb], 1u);
    } else if (qbucket) {
      qbucket[ql] = dir_base[l] + lo;
    }
  }
  unsigned long long c = count;
  for (int off = 32; off; off >>= 1) c += __shfl_xor(c, off);
  __shared__ unsigned long long s_c[4];
  if (lane_id() == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = s_c[0] + s_c[1] + s_c[2] + s_c[3];
    if (t) atomicAdd(cand_total, t);
  }
}

// Rare path: HashKey STRING equality (lsh.hpp:51-59) between the query's tuple and the tuple of
// the bucket with the same fingerprint.
__global__ __launch_bounds__(256) void hs_probe_slow_kernel(hs_tables_dev tabs,
                                                            const int32_t* __restrict__ qints,
                                                            int K, int L, uint32_t seed,
                                                            uint32_t* __restrict__ qstart,
                                                            uint32_t* __restrict__ qcount,
                                                            uint32_t* __restrict__ nslices,
                                                            uint64_t* __restrict__ cand_out,
                                                            unsigned long long* __restrict__ cand_total,
                                                            const uint32_t* __restrict__ slow,
                                                            const uint32_t* __restrict__ dir_base,
                                                            uint32_t nb_total,
                                                            uint32_t* __restrict__ bucket_count,
                                                            uint32_t* __restrict__ qbucket,
                                                            uint32_t* __restrict__ qrank) {
  const uint32_t total = slow[0];
  for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const uint32_t ql = slow[1 + e];
    const int l = (int)(ql % (uint32_t)L);
    int32_t t[HS_MAX_K], u[HS_MAX_K];
    for (int j = 0; j < K; ++j) t[j] = qints[(uint64_t)ql * K + j];
    const uint64_t key = hs_key_of(t, K, seed);
    const hs_table_dev& tb = tabs.t[l];
    uint32_t lo = 0, hi = tb.nb;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (tb.dir_key[mid] < key) lo = mid + 1; else hi = mid;
    }
    uint32_t gb = nb_total;
    if (lo < tb.nb && tb.dir_key[lo] == key) {
      for (int j = 0; j < K; ++j) u[j] = tb.dir_tuple[(uint64_t)lo * K + j];
      if (hs_key_equal(t, u, K)) {
        const uint32_t start = tb.dir_start[lo], count = tb.dir_start[lo + 1] - start;
        qstart[ql] = start;
        qcount[ql] = count;
        nslices[ql] = (count + HS_SLICE - 1) / HS_SLICE;
        if (cand_out) cand_out[ql] = count;
        atomicAdd(cand_total, (unsigned long long)count);
        if (dir_base) gb = dir_base[l] + lo;
      }
    }
    if (bucket_count) {
      qbucket[ql] = gb;
      qrank[ql] = atomicAdd(&bucket_count[gb], 1u);
    } else if (qbucket) {
      qbucket[ql] = gb;
    }
  }
}

// ---------------------------------------------------------------------------------------- qtables
// T[q][pos][aa] = sum_j (coord[aa][j] - c[q][8 pos + j])^2, fp64 then rounded to fp32.
// Rows are HS_TROW floats so a wave reads one with a single coalesced 128-byte load.
__global__ __launch_bounds__(256) void hs_qtables_kernel(const double* __restrict__ centers,
                                                         uint32_t nq, int k,
                                                         const double* __restrict__ coords,
                                                         int alphabet, float* __restrict__ tq) {
  // lane aa keeps its residue's 8 coordinates in registers; each group of 32 lanes walks rows
  // (q, pos) grid-stride: 8 broadcast loads of the centre's coordinates, one coalesced 128-B store
  const int aa = threadIdx.x & (HS_TROW - 1);
  double xc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) xc[j] = aa < alphabet ? coords[aa * 8 + j] : 0.0;
  const uint64_t rows = (uint64_t)nq * k;
  const uint64_t groups = (uint64_t)gridDim.x * (256 / HS_TROW);
  for (uint64_t row = (uint64_t)blockIdx.x * (256 / HS_TROW) + (threadIdx.x / HS_TROW); row < rows;
       row += groups) {
    const double* c = centers + row * 8;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double r = xc[j] - c[j];
      s += r * r;
    }
    tq[row * HS_TROW + aa] = aa < alphabet ? (float)s : 0.f;
  }
}

// ----------------------------------------------------------------------------------------- verify
// Work item = (query, table, slice of <= HS_SLICE bucket members), one wavefront each, taken
// grid-stride from the scanned slice counts.  The query's distance table lives in 25*PW VGPRs
// spread across lanes (lane aa holds T[pos][aa]); a candidate's squared distance is 25*PW
// ds_bpermute lookups + adds on one coalesced 16-byte load per lane.  Survivors of the fp32 filter
// (d2 <= R^2 (1 + 1e-5)) are compacted with a wave ballot; hs_finalize_kernel decides them exactly.
template <int PW, bool BRUTE>
__global__ __launch_bounds__(256) void hs_verify_kernel(hs_tables_dev tabs,
                                                        const uint4* __restrict__ brute_packed,
                                                        uint32_t brute_n,
                                                        const uint32_t* __restrict__ qstart,
                                                        const uint32_t* __restrict__ qcount,
                                                        const uint32_t* __restrict__ slice_off,
                                                        uint32_t nql, const float* __restrict__ tq,
                                                        int k, int L, float r2_hi,
                                                        uint32_t* __restrict__ prov_count,
                                                        uint32_t prov_cap, uint2* __restrict__ prov,
                                                        const float* __restrict__ q_thr,
                                                        float* __restrict__ slice_min) {
  const int lane = lane_id();
  const uint32_t waves_per_block = blockDim.x / WAVE;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * waves_per_block + (threadIdx.x >> 6));
  const uint32_t n_waves = gridDim.x * waves_per_block;
  const uint32_t total = BRUTE ? nql * ((brute_n + HS_SLICE - 1) / HS_SLICE) : slice_off[nql];
  for (uint32_t s = wave; s < total; s += n_waves) {
    uint32_t ql, sl, start, cnt;
    const uint4* packed;
    if (BRUTE) {
      const uint32_t per_q = (brute_n + HS_SLICE - 1) / HS_SLICE;
      ql = s / per_q;
      sl = s - ql * per_q;
      start = sl * HS_SLICE;
      cnt = min(HS_SLICE, brute_n - start);
      packed = brute_packed;
    } else {
      uint32_t lo = 0, hi = nql;  // largest ql with slice_off[ql] <= s
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (slice_off[mid] <= s) lo = mid; else hi = mid;
      }
      ql = lo;
      sl = s - slice_off[ql];
      start = qstart[ql] + sl * HS_SLICE;
      cnt = min(HS_SLICE, qcount[ql] - sl * HS_SLICE);
      packed = tabs.t[ql % (uint32_t)L].packed;
    }
    const uint32_t q = BRUTE ? ql : ql / (uint32_t)L;
    float T[25 * PW];
    const float* trow = tq + (uint64_t)q * k * HS_TROW + (lane & (HS_TROW - 1));
#pragma unroll
    for (int p = 0; p < 25 * PW; ++p) T[p] = p < k ? trow[p * HS_TROW] : 0.f;
    const uint32_t iters = (cnt + WAVE - 1) / WAVE;
    // brute-force top-k support: per-query thresholds, or a pure min pass over the slice
    const float thr = (BRUTE && q_thr) ? q_thr[q] : r2_hi;
    float run_min = INFINITY;
    for (uint32_t it = 0; it < iters; ++it) {
      const uint32_t i = it * WAVE + lane;
      const bool valid = i < cnt;
      const uint64_t pos = (uint64_t)start + (valid ? i : cnt - 1);
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < PW; ++w) {
        const uint4 pk = packed[pos * PW + w];
#define HS_LOOKUP(R) \
  sum += __int_as_float(__builtin_amdgcn_ds_bpermute(residue_x4<R>(pk), __float_as_int(T[25 * w + R])));
        HS_LOOKUP(0) HS_LOOKUP(1) HS_LOOKUP(2) HS_LOOKUP(3) HS_LOOKUP(4)
        HS_LOOKUP(5) HS_LOOKUP(6) HS_LOOKUP(7) HS_LOOKUP(8) HS_LOOKUP(9)
        HS_LOOKUP(10) HS_LOOKUP(11) HS_LOOKUP(12) HS_LOOKUP(13) HS_LOOKUP(14)
        HS_LOOKUP(15) HS_LOOKUP(16) HS_LOOKUP(17) HS_LOOKUP(18) HS_LOOKUP(19)
        HS_LOOKUP(20) HS_LOOKUP(21) HS_LOOKUP(22) HS_LOOKUP(23) HS_LOOKUP(24)
#undef HS_LOOKUP
      }
      if (BRUTE && slice_min) {
        run_min = fminf(run_min, valid ? sum : INFINITY);
        continue;
      }
      const bool pass = valid && sum <= thr;
      const unsigned long long m = __ballot(pass);
      if (m) {
        uint32_t base = 0;
        if (lane == 0) base = hs_reserve_survivors(prov_count, (uint32_t)__popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (pass) {
          const uint32_t idx = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          if (idx < prov_cap) prov[idx] = make_uint2(ql, (uint32_t)pos);
        }
      }
    }
    if (BRUTE && slice_min) {
      for (int off = 32; off; off >>= 1) run_min = fminf(run_min, __shfl_xor(run_min, off));
      if (lane == 0) slice_min[s] = run_min;  // s = q * slices_per_query + slice
    }
  }
}

// thr[q] = k-th smallest of the query's slice minima, widened by the fp32 filter band: at least k
// candidates (the slice minima themselves) pass it, and every true top-k member does (its fp32 sum
// is within 1.6e-6 relative of its exact d2).  Fewer than k slices -> +inf (small DBs: keep all).
__global__ __launch_bounds__(256) void hs_kth_min_kernel(const float* __restrict__ slice_min,
                                                         uint32_t per_q, uint32_t topk,
                                                         float* __restrict__ thr) {
  __shared__ float s_best[4];
  __shared__ uint32_t s_idx[4];
  const uint32_t q = blockIdx.x;
  const float* v = slice_min + (uint64_t)q * per_q;
  if (per_q < topk) {
    if (threadIdx.x == 0) thr[q] = INFINITY;
    return;
  }
  float last = -1.f;
  uint32_t last_idx = 0;
  bool first = true;
  for (uint32_t r = 0; r < topk; ++r) {
    // smallest (value, index) strictly greater than (last, last_idx)
    float best = INFINITY;
    uint32_t bi = 0xffffffffu;
    for (uint32_t i = threadIdx.x; i < per_q; i += 256) {
      const float x = v[i];
      const bool after = first || x > last || (x == last && i > last_idx);
      if (after && (x < best || (x == best && i < bi))) {
        best = x;
        bi = i;
      }
    }
    for (int off = 32; off; off >>= 1) {
      const float ob = __shfl_xor(best, off);
      const uint32_t oi = (uint32_t)__shfl_xor((int)bi, off);
      if (ob < best || (ob == best && oi < bi)) {
        best = ob;
        bi = oi;
      }
    }
    if (lane_id() == 0) {
      s_best[threadIdx.x >> 6] = best;
      s_idx[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    best = s_best[0];
    bi = s_idx[0];
    for (int w = 1; w < 4; ++w)
      if (s_best[w] < best || (s_best[w] == best && s_idx[w] < bi)) {
        best = s_best[w];
        bi = s_idx[w];
      }
    __syncthreads();
    last = best;
    last_idx = bi;
    first = false;
  }
  if (threadIdx.x == 0) thr[q] = last * (1.0f + 1e-5f) + 1e-30f;
}

// exact squared distance of every survivor of a top-k filter pass: key = (q, id), val = d2 bits
__global__ __launch_bounds__(256) void hs_topk_exact_kernel(const uint8_t* __restrict__ codes,
                                                            const double* __restrict__ centers,
                                                            const double* __restrict__ coords,
                                                            const uint2* __restrict__ prov,
                                                            const uint32_t* __restrict__ prov_count,
                                                            uint32_t prov_cap, int k,
                                                            uint64_t* __restrict__ out_key,
                                                            uint64_t* __restrict__ out_val) {
  const uint32_t n = min(*prov_count, prov_cap);
  for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const uint32_t q = prov[e].x, id = prov[e].y;
    const double d2 = exact_dist2(codes + (uint64_t)id * k, centers + (uint64_t)q * 8 * k, coords, k);
    out_key[e] = ((uint64_t)q << 37) | id;
    out_val[e] = (uint64_t)__double_as_longlong(d2);
  }
}

// --------------------------------------------------------------------------------------- finalize
// motif_both_points.cpp:176-183: d2 = sum_i (x_i - c_i)^2 left to right, x_i from the table.
__device__ __forceinline__ double exact_dist2(const uint8_t* __restrict__ row,
                                              const double* __restrict__ c,
                                              const double* __restrict__ coords, int k) {
  double d2 = 0.0;
  for (int p = 0; p < k; ++p) {
    const double* xc = coords + (int)row[p] * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double r = __dsub_rn(xc[j], c[8 * p + j]);
      d2 = __dadd_rn(d2, __dmul_rn(r, r));
    }
  }
  return d2;
}

// One thread per survivor of the fp32 filter: (1) first-seen dedupe -- the reference reports a DB
// id in the first table whose bucket holds it (label[] test, motif_both_points.cpp:233); ids are
// ascending inside a bucket, so membership in an earlier table's bucket is a binary search;
// (2) exact fp64 d2 and the reference's test d2 <= R*R (:239); (3) emit (q, table, id) key +
// sqrt(d2) (:241) for the ordering pass.
// SELF: the queries are indexed k-mers themselves (the self-join of Clustering()): their centre rows
// are rows of the coordinate table, taken from qcodes [nq][k] -- the same doubles an embedded centre
// row would hold, in the same operation order -- instead of a materialised [nq][8k] array.
template <bool SELF>
__global__ __launch_bounds__(256) void hs_finalize_kernel(hs_tables_dev tabs,
                                                          const uint8_t* __restrict__ codes,
                                                          const double* __restrict__ centers,
                                                          const uint8_t* __restrict__ qcodes,
                                                          const double* __restrict__ coords,
                                                          const uint32_t* __restrict__ qstart,
                                                          const uint32_t* __restrict__ qcount,
                                                          const uint2* __restrict__ prov,
                                                          const uint32_t* __restrict__ prov_count,
                                                          uint32_t prov_cap,
                                                          const uint32_t* __restrict__ sorted_ql,
                                                          int k, int L, double r2,
                                                          double r_sqrt, uint32_t q_base,
                                                          uint32_t self_first,
                                                          uint32_t* __restrict__ hit_count,
                                                          uint32_t hit_cap,
                                                          uint64_t* __restrict__ hit_key,
                                                          uint64_t* __restrict__ hit_val,
                                                          uint32_t* __restrict__ qcnt,
                                                          uint32_t* __restrict__ hit_rank) {
  // One wave = 64 survivors, one per lane.  The exact d2 is a serial fp64 chain per survivor, but
  // its inputs -- 8k doubles of the query's centre row -- are fetched by the WAVE: per position,
  // the 64 rows' 64-byte pieces go through LDS (4 lanes x 16 B per row: every byte fetched is
  // used), instead of 64 lanes walking 64 different rows 8 bytes at a time.
  constexpr int ROWB = 80;  // 64 B of a row + 16 B pad: conflict-free b128 reads
  __shared__ double s_coords[HS_ALPHABET_PAD * 8];
  __shared__ __attribute__((aligned(16))) unsigned char s_stage[4][64 * ROWB];
  __shared__ uint32_t s_q[4][64];
  __shared__ uint8_t s_code[4][64 * 76];  // the survivors' residue codes, row stride 76 (k <= 75)
  constexpr uint32_t HBUF = 128;          // hits a wave collects before it asks for global slots
  __shared__ uint64_t s_hk[4][HBUF], s_hv[4][HBUF];
  __shared__ uint32_t s_hr[4][HBUF];  // the hit's arrival number among its query's hits (with qcnt)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int t = tid; t < HS_ALPHABET_PAD * 8; t += 256) s_coords[t] = coords[t];
  __syncthreads();
  const uint32_t n = min(*prov_count, prov_cap);
  unsigned char* stage = s_stage[wave];
  const uint32_t wave_stride = gridDim.x * 4u * 64u;
  const int PW = (k + 24) / 25;  // packed words per k-mer (hs_packed_words)
  uint32_t n_buf = 0;  // wave-uniform: entries in the wave's hit buffer
  auto flush_hits = [&]() {
    if (!n_buf) return;
    uint32_t gbase = 0;
    if (lane == 0) gbase = atomicAdd(hit_count, n_buf);
    gbase = __shfl(gbase, 0);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = (uint32_t)lane; i < n_buf; i += 64u)
      if (gbase + i < hit_cap) {
        hit_key[gbase + i] = s_hk[wave][i];
        hit_val[gbase + i] = s_hv[wave][i];
        if (hit_rank) hit_rank[gbase + i] = s_hr[wave][i];
      }
    __builtin_amdgcn_wave_barrier();
    n_buf = 0;
  };
  for (uint32_t base = (blockIdx.x * 4u + (uint32_t)wave) * 64u; base < n; base += wave_stride) {
    const uint32_t e = base + (uint32_t)lane;
    uint32_t ql = e < n ? prov[e].x : 0xffffffffu;
    const uint32_t pos = e < n ? prov[e].y : 0u;
    const bool live = ql != 0xffffffffu;  // unused slot of a wave's reserved block (join kernels)
    // the join kernels name the probe by its position in segment order (no load in their hot loop)
    if (live && (ql & HS_PROV_INDIRECT)) ql = sorted_ql[ql & ~HS_PROV_INDIRECT];
    const uint32_t q = live ? ql / (uint32_t)L : 0u;
    const int l = live ? (int)(ql % (uint32_t)L) : 0;
    const uint32_t id = live ? tabs.t[l].ids[pos] : 0u;
    s_q[wave][lane] = q;
    __builtin_amdgcn_wave_barrier();
    {  // residue codes of this lane's survivor into LDS: from the table's bucket-ordered PACKED copy
       // (one 16-byte load per 25 residues at the survivor's position) rather than k byte loads from
       // the code array -- each of those touched 64 different cache lines per wave instruction
      uint8_t* dst = &s_code[wave][lane * 76];
      const uint4* pkp = tabs.t[l].packed + (uint64_t)pos * PW;
      for (int wd = 0; wd < PW; ++wd) {
        const uint4 pk = pkp[wd];
        const uint32_t w[5] = {pk.x, pk.y, pk.z, pk.w, 0u};
#pragma unroll
        for (int r = 0; r < 25; ++r) {
          const int bit = 5 * r, wi = bit >> 5, sh = bit & 31;
          uint32_t c = w[wi] >> sh;
          if (sh > 27) c |= w[wi + 1] << (32 - sh);
          const int p = 25 * wd + r;
          if (p < k) dst[p] = (uint8_t)(c & 31u);
        }
      }
      if constexpr (SELF) {
        const uint8_t* qc = qcodes + (uint64_t)q * k;
        uint8_t* qd = stage + lane * 76;  // SELF: no centre rows to stage, the area holds the queries' codes
        for (int p = 0; p < k; ++p) qd[p] = qc[p];
      }
    }
    // piece = 16 B: lane fetches pieces lane, lane + 64, ... of the wave's 64 x 64-byte block
    const int row0 = lane >> 2, part = lane & 3;
    const double* cbase = SELF ? s_coords : centers;  // SELF: no centre rows to stage (addresses unused)
    const uint64_t rs = SELF ? 0 : (uint64_t)8 * k;
    const double2* src0 = reinterpret_cast<const double2*>(cbase + (uint64_t)s_q[wave][row0] * rs) + part;
    const double2* src1 = reinterpret_cast<const double2*>(cbase + (uint64_t)s_q[wave][row0 + 16] * rs) + part;
    const double2* src2 = reinterpret_cast<const double2*>(cbase + (uint64_t)s_q[wave][row0 + 32] * rs) + part;
    const double2* src3 = reinterpret_cast<const double2*>(cbase + (uint64_t)s_q[wave][row0 + 48] * rs) + part;
    double2* const st0 = reinterpret_cast<double2*>(&stage[row0 * ROWB + part * 16]);
    double2* const st1 = reinterpret_cast<double2*>(&stage[(row0 + 16) * ROWB + part * 16]);
    double2* const st2 = reinterpret_cast<double2*>(&stage[(row0 + 32) * ROWB + part * 16]);
    double2* const st3 = reinterpret_cast<double2*>(&stage[(row0 + 48) * ROWB + part * 16]);
    // two positions in flight ahead of the one being summed (4 double2 per position of a row;
    // positions past the end re-read the last one: loads stay unconditional)
    double d2 = 0.0;
    if constexpr (SELF) {
      __builtin_amdgcn_wave_barrier();
      for (int p = 0; p < k; ++p) {
        // exact left-to-right fp64, one rounding per operation (PairwiseDistance hclust2.cpp:64-71)
        const double* xc = s_coords + (int)s_code[wave][lane * 76 + p] * 8;
        const double* cc = s_coords + (int)stage[lane * 76 + p] * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const double r = __dsub_rn(xc[j], cc[j]);
          d2 = __dadd_rn(d2, __dmul_rn(r, r));
        }
      }
    } else {
    // FOUR positions in flight ahead of the one being summed (the kernel is latency-bound: with two,
    // a wave spent ~ 23 k cycles on its 64 survivors at k = 15): four register sets, the loop unrolled
    // by four so that each set is a fixed group of registers; positions past the end re-read the last
    // one (loads stay unconditional) and are not summed.
    const int last = 4 * (k - 1);
#define HS_FIN_LOAD(A, B, C, D, O) \
  {                                \
    const int o_ = min((O), last); \
    A = src0[o_];                  \
    B = src1[o_];                  \
    C = src2[o_];                  \
    D = src3[o_];                  \
  }
    double2 a0, a1, a2, a3, b0, b1, b2, b3, e0, e1, e2, e3, g0, g1, g2, g3;
    HS_FIN_LOAD(a0, a1, a2, a3, 0)
    HS_FIN_LOAD(b0, b1, b2, b3, 4)
    HS_FIN_LOAD(e0, e1, e2, e3, 8)
    HS_FIN_LOAD(g0, g1, g2, g3, 12)
    // one position: stage this set's pieces, refill the set with the position four further on, sum
#define HS_FIN_STEP(A, B, C, D, P)                                                               \
  {                                                                                              \
    const int p = (P);                                                                           \
    *st0 = A;                                                                                    \
    *st1 = B;                                                                                    \
    *st2 = C;                                                                                    \
    *st3 = D;                                                                                    \
    HS_FIN_LOAD(A, B, C, D, 4 * (p + 4))                                                         \
    __builtin_amdgcn_wave_barrier();                                                             \
    double c[8];                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                              \
      const double2 v = *reinterpret_cast<const double2*>(&stage[lane * ROWB + j * 16]);         \
      c[2 * j] = v.x;                                                                            \
      c[2 * j + 1] = v.y;                                                                        \
    }                                                                                            \
    __builtin_amdgcn_wave_barrier();                                                             \
    if (p < k) { /* wave-uniform; exact left-to-right fp64, one rounding per operation           \
                    (PairwiseDistance_square :176-183) */                                        \
      const double* xc = s_coords + (int)s_code[wave][lane * 76 + p] * 8;                        \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                            \
        const double r = __dsub_rn(xc[j], c[j]);                                                 \
        d2 = __dadd_rn(d2, __dmul_rn(r, r));                                                     \
      }                                                                                          \
    }                                                                                            \
  }
    // FOUR positions in flight ahead of the one being summed (the kernel is latency-bound: with two,
    // a wave spent ~ 23 k cycles on its 64 survivors at k = 15): four register sets, the loop unrolled
    // by four; positions past the end re-read the last one (loads stay unconditional), not summed
    for (int p0 = 0; p0 < k; p0 += 4) {
      HS_FIN_STEP(a0, a1, a2, a3, p0)
      HS_FIN_STEP(b0, b1, b2, b3, p0 + 1)
      HS_FIN_STEP(e0, e1, e2, e3, p0 + 2)
      HS_FIN_STEP(g0, g1, g2, g3, p0 + 3)
    }
#undef HS_FIN_STEP
#undef HS_FIN_LOAD
    }
    // Search(): d2 <= R*R (motif_both_points.cpp:239); Clustering(): sqrt(d2) <= R
    // (hclust2.cpp:64-71,119-120), selected by a non-NaN r_sqrt.
    bool hit = live && ((r_sqrt == r_sqrt) ? (__dsqrt_rn(d2) <= r_sqrt) : (d2 <= r2));
    // the self-join drops a k-mer's pair with itself (Clustering() never compares a k-mer with
    // itself: its id is not in `centers` yet when it is visited, hclust2.cpp:116-131); an equal
    // k-mer under another id stays
    if (self_first != HS_NO_SELF && self_first + q_base + q == id) hit = false;
    // first-seen dedupe (label[], :233): the id was already reported if an EARLIER table's probed
    // bucket holds it, i.e. if its sorted position in that table falls inside the bucket's range
    // (one independent 4-byte load per earlier table)
    if (__ballot(hit && l > 0)) hit = hit && !seen_in_earlier_table(tabs, qstart, qcount, q, l, L, id, hit);
    // Hits go through a per-wave LDS buffer of HBUF entries and take their global slots when it
    // fills: same-address atomics complete at ~ 90 per microsecond on this part, and with hundreds of
    // hits per query (k = 15 at the C2 sizes: 3.3e6 wave iterations with hits) one counter access
    // per iteration WAS the kernel's time (37 of ~ 50 ms).
    const unsigned long long hm = __ballot(hit);
    if (hm) {
      const uint32_t cnt = (uint32_t)__popcll(hm);
      if (n_buf + cnt > HBUF) flush_hits();
      const uint32_t idx = n_buf + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
      // hits per query, and this hit's number among them: the ordering pass buckets by query without
      // another counter (hs_hit_place_kernel)
      const uint32_t rk = (hit && qcnt) ? atomicAdd(&qcnt[q], 1u) : 0u;
      if (hit) {
        s_hk[wave][idx] = ((uint64_t)(q_base + q) << 37) | ((uint64_t)l << 32) | id;
        s_hv[wave][idx] = (uint64_t)__double_as_longlong(__dsqrt_rn(d2));
        s_hr[wave][idx] = rk;
      }
      n_buf += cnt;
    }
  }
  flush_hits();
}

// hs_finalize_kernel for queries that are k-mers (qcodes given) and alphabets of up to HS_FIN_TABLE_ALPHABET
// residues: every term of the exact sum is then one of alphabet^2 x 8 values -- (x_r[j] - x_s[j])^2 for a
// member residue r meeting a query residue s, rounded once at the difference and once at the square as
// the reference does -- and the kernel looks them up (a 64-byte row of a table in LDS per position) and
// ADDS them in the reference's order: a third of the double-precision operations and half the LDS reads
// of fetching both residues' rows, no staging of 64-byte point pieces, no byte copies of residues (member
// and query are both 5-bit packed words -- the queries' by hs_pack_kernel -- read by constant shifts).  Hit-heavy batches spend their time
// here (k = 15 at the C2 sizes: 1.4e8 survivors per batch, 9.4 ms with the two-row form).
// (21: 80 x 21^2 = 35 KB of terms + 40 KB of per-wave hit buffers = 75 KB per workgroup, two of which share a
// CU's 160 KB; from 22 letters on only one would fit and the kernel -- which lives on waves in flight -- would
// lose half of them: those alphabets take the two-row form)
#define HS_FIN_TABLE_ALPHABET 21
// Workgroups of HS_FINC_WAVES waves: the kernel sits on s_waitcnt 87 % of its wave cycles (PMC) -- a chain of
// scattered loads per survivor -- so what it needs is waves in flight, and the 32 KB term table is per
// workgroup: 16 waves share one, two workgroups per CU = 32 waves (4 waves per table: 12).
#define HS_FINC_WAVES 16
__global__ __launch_bounds__(64 * HS_FINC_WAVES) void hs_finalize_codes_kernel(hs_tables_dev tabs,
                                                                const uint4* __restrict__ qpacked,
                                                                const double* __restrict__ coords, int alphabet,
                                                                const uint32_t* __restrict__ qstart,
                                                                const uint32_t* __restrict__ qcount,
                                                                const uint2* __restrict__ prov,
                                                                const uint32_t* __restrict__ prov_count,
                                                                uint32_t prov_cap,
                                                                const uint32_t* __restrict__ sorted_ql,
                                                                int k, int L, double r2, double r_sqrt,
                                                                uint32_t q_base, uint32_t self_first,
                                                                uint32_t* __restrict__ hit_count, uint32_t hit_cap,
                                                                uint64_t* __restrict__ hit_key,
                                                                uint64_t* __restrict__ hit_val,
                                                                uint32_t* __restrict__ qcnt,
                                                                uint32_t* __restrict__ hit_rank) {
  // [alphabet][alphabet] rows of 8 doubles at a stride of 10: with 64-byte rows the 16 lanes of one pass of a
  // 16-byte read meet in 4 bank groups (4-way conflicts on average), with 80-byte rows in 16
  extern __shared__ __attribute__((aligned(16))) double s_sq[];
  constexpr uint32_t HBUF = 128;
  __shared__ uint64_t s_hk[HS_FINC_WAVES][HBUF], s_hv[HS_FINC_WAVES][HBUF];
  __shared__ uint32_t s_hr[HS_FINC_WAVES][HBUF];  // the hit's arrival number among its query's hits (with qcnt)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < alphabet * alphabet * 8; e += 64 * HS_FINC_WAVES) {
    const int j = e & 7, rs = e >> 3, r = rs / alphabet, sq = rs - r * alphabet;
    const double d = __dsub_rn(coords[r * 8 + j], coords[sq * 8 + j]);
    s_sq[rs * 10 + j] = __dmul_rn(d, d);
  }
  __syncthreads();
  const uint32_t n = min(*prov_count, prov_cap);
  const uint32_t wave_stride = gridDim.x * (uint32_t)HS_FINC_WAVES * 64u;
  const int PW = (k + 24) / 25;
  uint32_t n_buf = 0;
  uint32_t pend_rk = 0;  // per lane: the answer of the lane's last counter access, not yet in the buffer ...
  int pend_idx = -1;     // ... and the buffer entry it belongs to
  auto flush_hits = [&]() {
    if (!n_buf) return;
    uint32_t gbase = 0;
    if (lane == 0) gbase = atomicAdd(hit_count, n_buf);
    gbase = __shfl(gbase, 0);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = (uint32_t)lane; i < n_buf; i += 64u)
      if (gbase + i < hit_cap) {
        hit_key[gbase + i] = s_hk[wave][i];
        hit_val[gbase + i] = s_hv[wave][i];
        if (hit_rank) hit_rank[gbase + i] = s_hr[wave][i];
      }
    __builtin_amdgcn_wave_barrier();
    n_buf = 0;
  };
  // A survivor's loads form a chain -- list entry, probe number (the join kernels name the probe by its
  // place in segment order), id and packed word, first-seen words: the first two links run ahead (the list
  // entry of the iteration after next and the probe number of the next one are requested before this
  // iteration's work).
  const uint2 dead = make_uint2(0xffffffffu, 0u);
  auto entry = [&](uint32_t b) { const uint32_t e_ = b + (uint32_t)lane; return (b < n && e_ < n) ? prov[e_] : dead; };
  auto probe_of = [&](const uint2 raw) {  // (index 0 for the lanes that need none: a load all the same, no branch)
    const bool ind = raw.x != 0xffffffffu && (raw.x & HS_PROV_INDIRECT);
    return sorted_ql ? sorted_ql[ind ? (raw.x & ~HS_PROV_INDIRECT) : 0u] : 0u;
  };
  const uint32_t base0 = (blockIdx.x * (uint32_t)HS_FINC_WAVES + (uint32_t)wave) * 64u;
  uint2 raw1 = entry(base0), raw2 = entry(base0 + wave_stride);
  uint32_t probe1 = probe_of(raw1);
  for (uint32_t base = base0; base < n; base += wave_stride) {
    const uint2 raw = raw1;
    const uint32_t probe = probe1;
    raw1 = raw2;
    probe1 = probe_of(raw1);
    raw2 = entry(base + 2u * wave_stride);
    const bool live = raw.x != 0xffffffffu;
    const uint32_t ql = (live && (raw.x & HS_PROV_INDIRECT)) ? probe : raw.x;
    const uint32_t pos = raw.y;
    const uint32_t q = live ? ql / (uint32_t)L : 0u;
    const int l = live ? (int)(ql % (uint32_t)L) : 0;
    const uint32_t id = live ? tabs.t[l].ids[pos] : 0u;
    const uint4* pkp = tabs.t[l].packed + (uint64_t)pos * PW;
    const uint4* qkp = qpacked + (uint64_t)q * PW;
    double d2 = 0.0;
    for (int wd = 0; wd < PW; ++wd) {
      const uint4 pk = pkp[wd], qk = qkp[wd];
      const uint32_t w[5] = {pk.x, pk.y, pk.z, pk.w, 0u}, wq[5] = {qk.x, qk.y, qk.z, qk.w, 0u};
#pragma unroll
      for (int r = 0; r < 25; ++r) {
        const int p = 25 * wd + r;
        if (p < k) {  // (wave-uniform)
          const int bit = 5 * r, wi = bit >> 5, sh = bit & 31;
          uint32_t c = w[wi] >> sh, cq = wq[wi] >> sh;
          if (sh > 27) {
            c |= w[wi + 1] << (32 - sh);
            cq |= wq[wi + 1] << (32 - sh);
          }
          c &= 31u;
          cq &= 31u;
          const double2* row = reinterpret_cast<const double2*>(s_sq + ((int)c * alphabet + (int)cq) * 10);
          // exact left-to-right sum of the rounded squares (PairwiseDistance_square :176-183)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const double2 v = row[j];
            d2 = __dadd_rn(d2, v.x);
            d2 = __dadd_rn(d2, v.y);
          }
        }
      }
    }
    bool hit = live && ((r_sqrt == r_sqrt) ? (__dsqrt_rn(d2) <= r_sqrt) : (d2 <= r2));
    if (self_first != HS_NO_SELF && self_first + q_base + q == id) hit = false;
    if (__ballot(hit && l > 0)) hit = hit && !seen_in_earlier_table(tabs, qstart, qcount, q, l, L, id, hit);
    const unsigned long long hm = __ballot(hit);
    if (hm) {
      // (the number the PREVIOUS hit of this lane drew from its query's counter goes to the buffer now: the
      // counter's answer has had a whole iteration to arrive)
      if (pend_idx >= 0) s_hr[wave][pend_idx] = pend_rk;
      pend_idx = -1;
      const uint32_t cnt = (uint32_t)__popcll(hm);
      if (n_buf + cnt > HBUF) flush_hits();
      const uint32_t idx = n_buf + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
      if (hit) {
        s_hk[wave][idx] = ((uint64_t)(q_base + q) << 37) | ((uint64_t)l << 32) | id;
        s_hv[wave][idx] = (uint64_t)__double_as_longlong(__dsqrt_rn(d2));
        if (qcnt) {
          pend_rk = atomicAdd(&qcnt[q], 1u);
          pend_idx = (int)idx;
        }
      }
      n_buf += cnt;
    }
  }
  if (pend_idx >= 0) s_hr[wave][pend_idx] = pend_rk;
  flush_hits();
}

// One thread per residue position: the position starts a window iff k residues of its own
// sequence follow it.  Neighbouring threads write neighbouring rows of the codes array.
__global__ __launch_bounds__(256) void hs_windows_kernel(const uint8_t* __restrict__ residues,
                                                         uint32_t n_residues,
                                                         const uint32_t* __restrict__ seq_start,
                                                         const uint32_t* __restrict__ win_off,
                                                         uint32_t n_seq, int k,
                                                         uint8_t* __restrict__ codes,
                                                         uint32_t* __restrict__ win_pos) {
  const uint32_t pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= n_residues) return;
  uint32_t lo = 0, hi = n_seq;  // largest s with seq_start[s] <= pos (empty sequences share a start:
  while (hi - lo > 1) {         // the last of them owns nothing either)
    const uint32_t mid = (lo + hi) >> 1;
    if (seq_start[mid] <= pos) lo = mid; else hi = mid;
  }
  const uint32_t end = seq_start[lo + 1];
  if (pos + (uint32_t)k > end) return;
  const uint32_t id = win_off[lo] + (pos - seq_start[lo]);
  win_pos[id] = pos;
  uint8_t* row = codes + (uint64_t)id * k;
  for (int p = 0; p < k; ++p) row[p] = residues[pos + p];
}

// ------------------------------------------------------------------------------------------ KLSH
// One wave per sequence: 512-bin histogram of its reduced-alphabet 3-mers in LDS (pcluster.cpp:
// 27-33), then lane i < bits runs the reference's serial dot product over the bins (lsh.cpp:8-15,
// product rounded, sum rounded) and tests cos(sum + b_i) + t_i >= 0 (lsh.cpp:44-46).
__global__ __launch_bounds__(256) void hs_klsh_kernel(const uint8_t* __restrict__ classes,
                                                      const uint64_t* __restrict__ seq_start,
                                                      uint64_t n_seq, const double* __restrict__ w,
                                                      const double* __restrict__ b,
                                                      const double* __restrict__ t, uint32_t bits,
                                                      uint64_t* __restrict__ codes,
                                                      uint64_t* __restrict__ uncertain) {
  __shared__ uint32_t s_hist[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* hist = s_hist[wave];
  for (uint64_t s = (uint64_t)blockIdx.x * 4 + wave; s < n_seq; s += (uint64_t)gridDim.x * 4) {
    const uint64_t lo = seq_start[s], len = seq_start[s + 1] - lo;
    for (int j = lane; j < 512; j += 64) hist[j] = 0;
    __builtin_amdgcn_wave_barrier();
    for (uint64_t i = lane; i + 3 <= len; i += 64) {
      const uint8_t* c = classes + lo + i;
      atomicAdd(&hist[(uint32_t)c[0] + 8u * c[1] + 64u * c[2]], 1u);  // Kmer2Integer util.hpp:244-250
    }
    __builtin_amdgcn_wave_barrier();
    bool bit = false, unc = false;
    if ((uint32_t)lane < bits) {
      const double* wi = w + (size_t)lane * 512;
      double sum = 0.0;
      for (int j = 0; j < 512; ++j) sum = __dadd_rn(sum, __dmul_rn((double)hist[j], wi[j]));
      sum = __dadd_rn(sum, b[lane]);
      const double v = __dadd_rn(cos(sum), t[lane]);
      bit = v >= 0.0;
      unc = fabs(v) < 1e-9;
    }
    const unsigned long long code = __ballot(bit), um = __ballot(unc);
    if (lane == 0) {
      codes[s] = len < 3 ? 0xffffffffffffffffull : code;
      if (uncertain) uncertain[s] = len < 3 ? 0ull : um;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Jump table over the sorted directory fingerprints: entry i fills the slots after its
// predecessor's up to its own; entry nb (one past the end) fills the rest.
__global__ __launch_bounds__(256) void hs_dir_jump_kernel(const uint64_t* __restrict__ key, uint32_t nb,
                                                          uint32_t shift, uint32_t n_slots,
                                                          uint32_t* __restrict__ jump) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i > nb) return;
  const uint32_t cur = i < nb ? (uint32_t)(key[i] >> shift) : n_slots;
  const uint32_t first = i ? (uint32_t)(key[i - 1] >> shift) + 1 : 0;
  for (uint32_t t = first; t <= cur; ++t) jump[t] = (uint32_t)i;
}

// Directory records: what a probe needs of a bucket -- fingerprint, boundaries, the tuple of the exact check --
// in one 64-byte line instead of four arrays (fingerprints, boundaries, 4 K bytes of tuple: 5-6 scattered
// 64-byte sectors per probe, the probe kernel's whole cost at configs[2]'s 3.2e7 probes per batch).
__global__ __launch_bounds__(256) void hs_dir_records_kernel(const uint64_t* __restrict__ key,
                                                             const uint32_t* __restrict__ start,
                                                             const int32_t* __restrict__ tuple, uint32_t nb, int K,
                                                             uint4* __restrict__ rec, uint32_t* __restrict__ flag) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b >= nb) return;
  const uint64_t kb = key[b];
  const uint32_t s0 = start[b];
  uint32_t w[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) w[j] = 0;
  bool wide = false;
  const int32_t* t = tuple + (uint64_t)b * K;
#pragma unroll
  for (int j = 0; j < HS_REC_MAX_K; ++j)
    if (j < K) {
      const int32_t v = t[j];
      wide = wide || v != (int32_t)(int16_t)v;
      w[j >> 1] |= ((uint32_t)v & 0xffffu) << (16 * (j & 1));
    }
  if (wide) atomicOr(flag, 1u);
  uint4* r = rec + 4 * (uint64_t)b;
  r[0] = make_uint4((uint32_t)kb, (uint32_t)(kb >> 32), s0, start[b + 1] - s0);
  r[1] = make_uint4(w[0], w[1], w[2], w[3]);
  r[2] = make_uint4(w[4], w[5], w[6], w[7]);
  r[3] = make_uint4(w[8], w[9], w[10], w[11]);
}

__global__ __launch_bounds__(256) void hs_invert_perm_kernel(const uint32_t* __restrict__ perm,
                                                             uint32_t n, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[perm[i]] = i;
}

// hs_index_load: a table read from a file is checked before any kernel indexes with it.
// flag bits: 1 id out of range, 2 id twice (ids is not a permutation), 4 bucket boundaries not
// strictly ascending from 0 to n, 8 fingerprints not strictly ascending, 16 a bucket's tuple does
// not have its fingerprint, 32 ids not ascending inside a bucket (the reference appends ids in
// ascending order, motif_both_points.cpp:212).  pos_of must be filled with 0xffffffff beforehand;
// it comes out as the inverse permutation (what hs_invert_perm_kernel writes for a trusted table).
__global__ __launch_bounds__(256) void hs_validate_ids_kernel(const uint32_t* __restrict__ ids, uint32_t n,
                                                              uint32_t* __restrict__ pos_of,
                                                              uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t id = ids[i];
  if (id >= n) {
    atomicOr(flag, 1u);
    return;
  }
  if (atomicExch(&pos_of[id], i) != 0xffffffffu) atomicOr(flag, 2u);
}

__global__ __launch_bounds__(256) void hs_validate_dir_kernel(const uint32_t* __restrict__ dir_start,
                                                              const uint64_t* __restrict__ dir_key,
                                                              const int32_t* __restrict__ dir_tuple,
                                                              const uint32_t* __restrict__ ids,
                                                              uint32_t nb, uint32_t n, int K, uint32_t seed,
                                                              uint32_t* __restrict__ flag,
                                                              uint32_t* __restrict__ max_bucket) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b == 0 && ((nb ? dir_start[0] : 0u) != 0u || dir_start[nb] != n || (n != 0u) != (nb != 0u)))
    atomicOr(flag, 4u);
  if (b >= nb) return;
  const uint32_t lo = dir_start[b], hi = dir_start[b + 1];
  if (!(lo < hi) || hi > n) {
    atomicOr(flag, 4u);
    return;
  }
  if (b + 1 < nb && !(dir_key[b] < dir_key[b + 1])) atomicOr(flag, 8u);
  int32_t t[HS_MAX_K];
  for (int i = 0; i < K; ++i) t[i] = dir_tuple[(uint64_t)b * K + i];
  if (hs_key_of(t, K, seed) != dir_key[b]) atomicOr(flag, 16u);
  atomicMax(max_bucket, hi - lo);
}

// one thread per entry: ids may only descend where a bucket starts (binary search on descents only)
__global__ __launch_bounds__(256) void hs_validate_order_kernel(const uint32_t* __restrict__ ids,
                                                                const uint32_t* __restrict__ dir_start,
                                                                uint32_t nb, uint32_t n,
                                                                uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x + 1;
  if (i >= n) return;
  if (ids[i - 1] < ids[i]) return;
  uint32_t lo = 0, hi = nb;  // is i one of dir_start[0 .. nb)?
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (dir_start[mid] < i) lo = mid + 1; else hi = mid;
  }
  if (lo >= nb || dir_start[lo] != i) atomicOr(flag, 32u);
}

// Brute force (motif_both_points_noLSH.cpp:27-34,44-50): sqrt form, hit iff !(dis > R).
__global__ __launch_bounds__(256) void hs_bf_finalize_kernel(const uint8_t* __restrict__ codes,
                                                             const double* __restrict__ centers,
                                                             const double* __restrict__ coords,
                                                             const uint2* __restrict__ prov,
                                                             const uint32_t* __restrict__ prov_count,
                                                             uint32_t prov_cap, int k, double R,
                                                             uint32_t q_base,
                                                             uint32_t* __restrict__ hit_count,
                                                             uint32_t hit_cap,
                                                             uint64_t* __restrict__ hit_key,
                                                             uint64_t* __restrict__ hit_val) {
  const uint32_t n = min(*prov_count, prov_cap);
  for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const uint32_t q = prov[e].x, id = prov[e].y;
    const double d2 = exact_dist2(codes + (uint64_t)id * k, centers + (uint64_t)q * 8 * k, coords, k);
    const double dis = __dsqrt_rn(d2);
    if (!(dis > R)) {
      const uint32_t idx = atomicAdd(hit_count, 1u);
      if (idx < hit_cap) {
        hit_key[idx] = ((uint64_t)(q_base + q) << 37) | id;
        hit_val[idx] = (uint64_t)__double_as_longlong(dis);
      }
    }
  }
}

// Ordering of a batch's hits without a sort over the whole list and without the host knowing their
// number: hits are bucketed by query (qoff = exclusive scan of the per-query counts hs_finalize_kernel
// kept), then every query orders its own hits by (table of first sight, id) and writes them out -- the
// reference's file order (motif_both_points.cpp:224-245).  A query with up to HS_ORDER_MAX hits is
// ordered by one thread (insertion sort); one with more goes on one of two lists (up to 1024 / up to
// HS_ORDER_BLOCK_MAX hits) whose queries are ordered by a block each, with a bitonic sort in LDS
// (hs_hit_order_block_kernel): short k-mers at a loose radius have hundreds of hits per query (k = 15 at
// the C2 sizes: 545 on average), and the radix sort over the whole list those batches fell back to cost
// as much as their join.  A query with more than HS_ORDER_BLOCK_MAX hits goes on a third list: a block of
// 1024 threads sorts it in chunks of that many words in LDS and merges the chunks through global memory
// (hs_hit_order_huge_kernel).  Only a query with 2^27 hits or more -- the place no longer fits beside
// (table, id) in a word -- raises *big (the caller then falls back to the sort over the whole list).
#define HS_ORDER_MAX 48u
#define HS_ORDER_BLOCK_MAX 8192u
#define HS_ORDER_LIST_HEAD 8u   // words in front of the three lists: [0..2] entries, [3..5] work counters
__global__ __launch_bounds__(256) void hs_hit_place_kernel(const uint64_t* __restrict__ key,
                                                           const uint64_t* __restrict__ val,
                                                           const uint32_t* __restrict__ hit_count,
                                                           uint32_t hit_cap, uint32_t q_base,
                                                           const uint32_t* __restrict__ qoff,
                                                           const uint32_t* __restrict__ rank,
                                                           ulonglong2* __restrict__ kv) {
  const uint32_t n = min(*hit_count, hit_cap);
  for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const uint64_t kk = key[e];
    const uint32_t q = (uint32_t)(kk >> 37) - q_base;
    const uint32_t slot = qoff[q] + rank[e];  // (the hit's number among its query's: hs_finalize_kernel)
    if (slot < hit_cap) {
      kv[slot] = make_ulonglong2(kk, val[e]);  // (key and value side by side: one scattered 16-byte store)
    }
  }
}

// qlist: HS_ORDER_LIST_HEAD words, then the three lists (nq entries each): queries of up to 1024 hits, of
// up to HS_ORDER_BLOCK_MAX, of more
__global__ __launch_bounds__(256) void hs_hit_order_kernel(const uint32_t* __restrict__ qoff, uint32_t nq,
                                                           const uint32_t* __restrict__ hit_count,
                                                           uint32_t hit_cap, ulonglong2* __restrict__ kv,
                                                           uint32_t* __restrict__ big,
                                                           uint32_t* __restrict__ qlist,
                                                           uint32_t* __restrict__ out_q,
                                                           uint32_t* __restrict__ out_id,
                                                           uint32_t* __restrict__ out_table,
                                                           double* __restrict__ out_dist, uint64_t out_room) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (*hit_count > hit_cap) return;  // the batch is repeated with larger buffers anyway
  uint32_t lo = 0, m = 0;
  if (q < nq) {
    lo = qoff[q];
    m = qoff[q + 1] - lo;
  }
  if (m >= (1u << 27)) {
    atomicOr(big, 1u);
    m = 0;
  }
  {  // the queries a block orders go on their lists: one counter access per wave and list (one per query --
     // 7.7e4 same-address atomics at ~ 90 per microsecond -- was this kernel's whole time in a hit-heavy batch)
    const int which = m > HS_ORDER_BLOCK_MAX ? 2 : m > 1024u ? 1 : m > HS_ORDER_MAX ? 0 : -1;
    const int lane = (int)(threadIdx.x & 63);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const unsigned long long mm = __ballot(which == c);
      if (!mm) continue;
      uint32_t base = 0;
      if (lane == (int)__builtin_ctzll(mm)) base = atomicAdd(&qlist[c], (uint32_t)__popcll(mm));
      base = __shfl(base, (int)__builtin_ctzll(mm));
      if (which == c)
        qlist[HS_ORDER_LIST_HEAD + (size_t)c * nq + base + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull))] = q;
    }
    if (which >= 0) return;
  }
  if (!m) return;
  // insertion sort of the query's hits by key (distinct: (table, id) is unique per query)
  for (uint32_t i = 1; i < m; ++i) {
    const ulonglong2 cur = kv[lo + i];
    const uint64_t kk = cur.x;
    uint32_t j = i;
    while (j > 0 && kv[lo + j - 1].x > kk) {
      kv[lo + j] = kv[lo + j - 1];
      --j;
    }
    kv[lo + j] = cur;
  }
  for (uint32_t i = 0; i < m; ++i) {
    const uint64_t o = (uint64_t)lo + i;
    if (o >= out_room) break;
    const uint64_t kk = kv[lo + i].x;
    out_q[o] = (uint32_t)(kk >> 37);
    if (out_table) out_table[o] = (uint32_t)((kk >> 32) & 31u);
    out_id[o] = (uint32_t)kk;
    out_dist[o] = __longlong_as_double((long long)kv[lo + i].y);
  }
}

// One block per listed query: its hits' (table, id) -- 37 bits -- with the hit's place in the query's
// range below them -- 27 bits -- sorted as one 64-bit word in LDS, padded with all-ones words to a
// power of two; the distance follows through the place.  CAP = 1024 or HS_ORDER_BLOCK_MAX words of LDS,
// NT = 256 or 1024 threads.
template <uint32_t CAP, uint32_t NT>
__global__ __launch_bounds__(NT) void hs_hit_order_block_kernel(const uint32_t* __restrict__ qoff, uint32_t nq,
                                                                 const uint32_t* __restrict__ hit_count,
                                                                 uint32_t hit_cap, uint32_t which,
                                                                 const ulonglong2* __restrict__ kv,
                                                                 uint32_t* __restrict__ qlist,
                                                                 uint32_t* __restrict__ out_q,
                                                                 uint32_t* __restrict__ out_id,
                                                                 uint32_t* __restrict__ out_table,
                                                                 double* __restrict__ out_dist, uint64_t out_room) {
  __shared__ uint64_t sk[CAP];
  if (*hit_count > hit_cap) return;
  const uint32_t n_list = qlist[which];
  const uint32_t* const list = qlist + HS_ORDER_LIST_HEAD + (size_t)which * nq;
  const uint32_t tid = threadIdx.x;
  // The short queries are dealt to the blocks in turn (a shared work counter is one same-address atomic per
  // query, ~ 90 per microsecond: with 6e4 of them it showed); the long ones, few and of very different
  // lengths, are drawn from a counter.
  for (uint32_t at = blockIdx.x;; at += gridDim.x) {
    __syncthreads();  // (the previous query's words have been written out)
    if (CAP > 1024u) {
      if (tid == 0) sk[0] = atomicAdd(&qlist[3 + which], 1u);  // the block's next query, by way of word 0
      __syncthreads();
      at = (uint32_t)sk[0];
      __syncthreads();
    }
    if (at >= n_list) return;
    const uint32_t q = list[at];
    const uint32_t lo = qoff[q], m = qoff[q + 1] - lo;
    uint32_t P = 2u * NT;  // (every thread has a pair in every step)
    while (P < m) P <<= 1;
    for (uint32_t i = tid; i < P; i += NT)
      sk[i] = i < m ? ((kv[lo + i].x & ((1ull << 37) - 1ull)) << 27) | (uint64_t)i : ~0ull;
    __syncthreads();
    for (uint32_t size = 2u; size <= P; size <<= 1) {
      for (uint32_t stride = size >> 1; stride > 0u; stride >>= 1) {
        for (uint32_t t = tid; t < (P >> 1); t += NT) {
          const uint32_t i = 2u * t - (t & (stride - 1u));  // the pair's lower index
          const uint32_t j = i + stride;
          const bool up = (i & size) == 0u;
          const uint64_t a = sk[i], b = sk[j];
          if ((a > b) == up) {
            sk[i] = b;
            sk[j] = a;
          }
        }
        __syncthreads();
      }
    }
    const uint32_t q_abs = (uint32_t)(kv[lo].x >> 37);
    for (uint32_t i = tid; i < m; i += NT) {
      const uint64_t o = (uint64_t)lo + i;
      if (o >= out_room) break;
      const uint64_t e = sk[i];
      const uint64_t ti = e >> 27;
      out_q[o] = q_abs;
      if (out_table) out_table[o] = (uint32_t)(ti >> 32) & 31u;
      out_id[o] = (uint32_t)ti;
      out_dist[o] = __longlong_as_double((long long)kv[lo + (uint32_t)(e & ((1u << 27) - 1u))].y);
    }
  }
}

// Sorting network used below: every comparator ascending (the smaller word to the lower index).  A merge
// of two sorted runs of size / 2 first compares i with its MIRROR image in the run pair, then halves the
// stride down to 1.  With all comparators ascending, words past the end of the list behave as +infinity
// without being stored: a comparator whose upper index is past the end does nothing.
__device__ __forceinline__ void order_lds_steps(uint64_t* sk, uint32_t n_words, uint32_t size, bool mirror_first,
                                                uint32_t first_stride, uint32_t tid, uint32_t n_threads) {
  if (mirror_first) {
    const uint32_t half = size >> 1;
    for (uint32_t t = tid; t < (n_words >> 1); t += n_threads) {
      const uint32_t blk = t / half, off = t - blk * half;
      const uint32_t i = blk * size + off, j = blk * size + size - 1u - off;
      const uint64_t a = sk[i], b = sk[j];
      if (a > b) {
        sk[i] = b;
        sk[j] = a;
      }
    }
    __syncthreads();
  }
  for (uint32_t stride = first_stride; stride > 0u; stride >>= 1) {
    for (uint32_t t = tid; t < (n_words >> 1); t += n_threads) {
      const uint32_t i = 2u * t - (t & (stride - 1u)), j = i + stride;
      const uint64_t a = sk[i], b = sk[j];
      if (a > b) {
        sk[i] = b;
        sk[j] = a;
      }
    }
    __syncthreads();
  }
}

// One block of 1024 threads per query of the third list.  scratch: as many words as the hit lists hold (the
// unbucketed list's keys, free once hs_hit_place_kernel has run); the query's words live at its own
// range [lo, lo + m) of it.  Chunks of HS_ORDER_BLOCK_MAX words are sorted in LDS; merges of larger runs do
// their mirror step and their strides >= a chunk in global memory (volatile accesses: the block's own
// stores must be what its later loads see), the rest chunk by chunk in LDS again.
__global__ __launch_bounds__(1024) void hs_hit_order_huge_kernel(const uint32_t* __restrict__ qoff, uint32_t nq,
                                                                 const uint32_t* __restrict__ hit_count,
                                                                 uint32_t hit_cap, const uint32_t* __restrict__ big,
                                                                 const ulonglong2* __restrict__ kv,
                                                                 uint32_t* __restrict__ qlist,
                                                                 uint64_t* scratch,
                                                                 uint32_t* __restrict__ out_q,
                                                                 uint32_t* __restrict__ out_id,
                                                                 uint32_t* __restrict__ out_table,
                                                                 double* __restrict__ out_dist, uint64_t out_room) {
  constexpr uint32_t CH = HS_ORDER_BLOCK_MAX;
  __shared__ uint64_t sk[CH];
  if (*hit_count > hit_cap || *big) return;  // (*big: the whole list is about to be sorted from `scratch`)
  const uint32_t n_list = qlist[2];
  const uint32_t* const list = qlist + HS_ORDER_LIST_HEAD + 2u * (size_t)nq;
  const uint32_t tid = threadIdx.x;
  for (;;) {
    __syncthreads();
    if (tid == 0) sk[0] = atomicAdd(&qlist[5], 1u);
    __syncthreads();
    const uint32_t at = (uint32_t)sk[0];
    __syncthreads();
    if (at >= n_list) return;
    const uint32_t q = list[at];
    const uint32_t lo = qoff[q], m = qoff[q + 1] - lo;
    volatile uint64_t* const G = scratch + lo;
    uint32_t P = CH;
    while (P < m) P <<= 1;
    // sorted chunks
    for (uint32_t base = 0; base < m; base += CH) {
      for (uint32_t i = tid; i < CH; i += 1024u)
        sk[i] = base + i < m ? ((kv[lo + base + i].x & ((1ull << 37) - 1ull)) << 27) | (uint64_t)(base + i) : ~0ull;
      __syncthreads();
      for (uint32_t size = 2u; size <= CH; size <<= 1) order_lds_steps(sk, CH, size, true, size >> 2, tid, 1024u);
      for (uint32_t i = tid; i < CH && base + i < m; i += 1024u) G[base + i] = sk[i];
      __syncthreads();
    }
    // merges of runs of a chunk and more
    for (uint32_t size = 2u * CH; size <= P; size <<= 1) {
      const uint32_t half = size >> 1;
      for (uint32_t t = tid; t < (P >> 1); t += 1024u) {
        const uint32_t blk = t / half, off = t - blk * half;
        const uint32_t i = blk * size + off, j = blk * size + size - 1u - off;
        if (j < m) {
          const uint64_t a = G[i], b = G[j];
          if (a > b) {
            G[i] = b;
            G[j] = a;
          }
        }
      }
      __syncthreads();
      for (uint32_t stride = size >> 2; stride >= CH; stride >>= 1) {
        for (uint32_t t = tid; t < (P >> 1); t += 1024u) {
          const uint32_t i = 2u * t - (t & (stride - 1u)), j = i + stride;
          if (j < m) {
            const uint64_t a = G[i], b = G[j];
            if (a > b) {
              G[i] = b;
              G[j] = a;
            }
          }
        }
        __syncthreads();
      }
      for (uint32_t base = 0; base < m; base += CH) {
        for (uint32_t i = tid; i < CH; i += 1024u) sk[i] = base + i < m ? G[base + i] : ~0ull;
        __syncthreads();
        order_lds_steps(sk, CH, CH, false, CH >> 1, tid, 1024u);
        for (uint32_t i = tid; i < CH && base + i < m; i += 1024u) G[base + i] = sk[i];
        __syncthreads();
      }
    }
    const uint32_t q_abs = (uint32_t)(kv[lo].x >> 37);
    for (uint32_t i = tid; i < m; i += 1024u) {
      const uint64_t o = (uint64_t)lo + i;
      if (o >= out_room) break;
      const uint64_t e = G[i];
      const uint64_t ti = e >> 27;
      out_q[o] = q_abs;
      if (out_table) out_table[o] = (uint32_t)(ti >> 32) & 31u;
      out_id[o] = (uint32_t)ti;
      out_dist[o] = __longlong_as_double((long long)kv[lo + (uint32_t)(e & ((1u << 27) - 1u))].y);
    }
  }
}

__global__ __launch_bounds__(256) void hs_unpack_hits_kernel(const uint64_t* __restrict__ key,
                                                             const uint64_t* __restrict__ val,
                                                             uint32_t n, uint32_t* __restrict__ q,
                                                             uint32_t* __restrict__ id,
                                                             uint32_t* __restrict__ table,
                                                             double* __restrict__ dist) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t kk = key[i];
  q[i] = (uint32_t)(kk >> 37);
  if (table) table[i] = (uint32_t)((kk >> 32) & 31u);
  id[i] = (uint32_t)kk;
  dist[i] = __longlong_as_double((long long)val[i]);
}

// ---- merge of the table-partitioned layout (hs_merge_first_table_dev) --------------------------------
// Every rank holds a block of the L tables over ALL k-mers and answers ALL queries; gathered, one (query, id)
// pair appears once per rank whose tables hold it in the query's bucket, each time with the smallest of THAT
// rank's tables.  The reference reports an id in the first table whose probed bucket holds it
// (motif_both_points.cpp:232-238) -- the smallest table over all ranks -- so: order the tuples by
// (q, id, table), keep the first of every (q, id) run, order what is kept by (q, table, id).
__global__ __launch_bounds__(256) void hs_merge_key1_kernel(const uint32_t* __restrict__ q, const uint32_t* __restrict__ id,
                                                            const uint32_t* __restrict__ table,
                                                            const double* __restrict__ dist, uint32_t n,
                                                            uint64_t* __restrict__ key, uint64_t* __restrict__ val) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  key[i] = ((uint64_t)q[i] << 37) | ((uint64_t)id[i] << 5) | (uint64_t)(table[i] & 31u);
  val[i] = (uint64_t)__double_as_longlong(dist[i]);
}
// flag[i] = 1 where a (q, id) run starts; flag[n] = 0 closes the scan
__global__ __launch_bounds__(256) void hs_merge_flag_kernel(const uint64_t* __restrict__ key, uint32_t n,
                                                            uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i > n) return;
  flag[i] = (i < n && (i == 0 || (key[i] >> 5) != (key[i - 1] >> 5))) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void hs_merge_compact_kernel(const uint64_t* __restrict__ key,
                                                               const uint64_t* __restrict__ val,
                                                               const uint32_t* __restrict__ pos, uint32_t n,
                                                               uint64_t* __restrict__ key2, uint64_t* __restrict__ val2) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || pos[i + 1] == pos[i]) return;
  const uint64_t kk = key[i];
  const uint64_t q = kk >> 37, id = (kk >> 5) & 0xffffffffull, t = kk & 31ull;
  key2[pos[i]] = (q << 37) | (t << 32) | id;
  val2[pos[i]] = val[i];
}

inline unsigned blocks_for(uint64_t n, unsigned per = 256) { return (unsigned)((n + per - 1) / per); }

}  // namespace

// ================================================================================= launchers
hipError_t hs_launch_merge_key1(const uint32_t* d_q, const uint32_t* d_id, const uint32_t* d_table, const double* d_dist,
                                uint32_t n, uint64_t* d_key, uint64_t* d_val, hipStream_t s) {
  if (!n) return hipSuccess;
  hs_merge_key1_kernel<<<blocks_for(n), 256, 0, s>>>(d_q, d_id, d_table, d_dist, n, d_key, d_val);
  return hipGetLastError();
}
hipError_t hs_launch_merge_flag(const uint64_t* d_key, uint32_t n, uint32_t* d_flag, hipStream_t s) {
  hs_merge_flag_kernel<<<blocks_for((uint64_t)n + 1), 256, 0, s>>>(d_key, n, d_flag);
  return hipGetLastError();
}
hipError_t hs_launch_merge_compact(const uint64_t* d_key, const uint64_t* d_val, const uint32_t* d_pos, uint32_t n,
                                   uint64_t* d_key2, uint64_t* d_val2, hipStream_t s) {
  if (!n) return hipSuccess;
  hs_merge_compact_kernel<<<blocks_for(n), 256, 0, s>>>(d_key, d_val, d_pos, n, d_key2, d_val2);
  return hipGetLastError();
}

hipError_t hs_launch_embed(const uint8_t* d_codes, uint64_t n, int k, const double* d_coords,
                           double* d_out, hipStream_t s) {
  const uint64_t total = n * 8ull * k;
  if (!total) return hipSuccess;
  unsigned blocks = (unsigned)std::min<uint64_t>(blocks_for(total), 256u * 16u);
  hs_embed_kernel<<<blocks, 256, 0, s>>>(d_codes, total, d_coords, d_out);
  return hipGetLastError();
}

template <bool FROM_CODES>
static hipError_t launch_hash(const uint8_t* d_codes, const double* d_pts, uint64_t n, int k,
                              const double* d_aT, int ldf, const double* d_b, int F, double W,
                              const double* d_coords, int32_t* d_out, int out_stride, hipStream_t s) {
  if (!n || !F) return hipSuccess;
  const unsigned blocks = blocks_for(n);
  const size_t lds = FROM_CODES ? 256u * (size_t)k : 0;
  const int kc = F % 16 == 0 ? 16 : F % 8 == 0 ? 8 : F % 5 == 0 ? 5 : F % 4 == 0 ? 4 : F % 2 == 0 ? 2 : 1;
  const unsigned chunks = (unsigned)(F / kc);
  // enough block rows to reach ~16 blocks per CU when the point count alone does not
  const unsigned gy = std::max(1u, std::min(chunks, 4096u / std::max(blocks, 1u)));
  const dim3 grid(blocks, gy);
#define HS_HASH(KC)                                                                            \
  hs_hash_kernel<KC, FROM_CODES><<<grid, 256, lds, s>>>(d_codes, d_pts, n, k, d_aT, ldf, d_b, F, W, \
                                                        d_coords, d_out, out_stride)
  if (F % 16 == 0) HS_HASH(16);
  else if (F % 8 == 0) HS_HASH(8);
  else if (F % 5 == 0) HS_HASH(5);
  else if (F % 4 == 0) HS_HASH(4);
  else if (F % 2 == 0) HS_HASH(2);
  else HS_HASH(1);
#undef HS_HASH
  return hipGetLastError();
}

hipError_t hs_launch_hash_codes(const uint8_t* d_codes, uint64_t n, int k, const double* d_aT, int ldf,
                                const double* d_b, int F, double W, const double* d_coords,
                                int32_t* d_out, int out_stride, hipStream_t s) {
  return launch_hash<true>(d_codes, nullptr, n, k, d_aT, ldf, d_b, F, W, d_coords, d_out, out_stride, s);
}
hipError_t hs_launch_hash_points(const double* d_pts, uint64_t n, int k, const double* d_aT, int ldf,
                                 const double* d_b, int F, double W, int32_t* d_out, int out_stride,
                                 hipStream_t s) {
  return launch_hash<false>(nullptr, d_pts, n, k, d_aT, ldf, d_b, F, W, nullptr, d_out, out_stride, s);
}

__global__ __launch_bounds__(256) void hs_transpose_f64_kernel(const double* __restrict__ in, int rows,
                                                               int cols, double* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (uint64_t)rows * cols) return;
  const int r = (int)(t / cols), c = (int)(t % cols);
  out[(uint64_t)c * rows + r] = in[t];
}
hipError_t hs_launch_transpose_f64(const double* d_in, int rows, int cols, double* d_out, hipStream_t s) {
  if (!rows || !cols) return hipSuccess;
  hs_transpose_f64_kernel<<<blocks_for((uint64_t)rows * cols), 256, 0, s>>>(d_in, rows, cols, d_out);
  return hipGetLastError();
}

hipError_t hs_launch_keys(const int32_t* d_ints, uint64_t n, int stride, int K, uint32_t seed,
                          uint64_t* d_keys, uint32_t* d_ids, hipStream_t s) {
  if (!n) return hipSuccess;
  hs_keys_kernel<<<blocks_for(n), 256, 0, s>>>(d_ints, n, stride, K, seed, d_keys, d_ids);
  return hipGetLastError();
}

hipError_t hs_launch_check_runs(const uint64_t* d_keys_sorted, const uint32_t* d_ids_sorted,
                                const int32_t* d_ints, uint64_t n, int K, uint32_t* d_flag,
                                uint32_t* d_slow, uint32_t slow_cap, bool exhaustive, int sorted_from_bit,
                                hipStream_t s) {
  if (n < 2) return hipSuccess;
  if (exhaustive) {
    hs_check_runs_slow_kernel<<<2048, 256, 0, s>>>(d_ids_sorted, d_ints, K, d_slow, slow_cap, n,
                                                   d_keys_sorted, d_flag);
    return hipGetLastError();
  }
  hipError_t e = hipMemsetAsync(d_slow, 0, 4, s);
  if (e != hipSuccess) return e;
  hs_check_runs_kernel<<<blocks_for(n - 1), 256, 0, s>>>(d_keys_sorted, d_ids_sorted, d_ints, n, K,
                                                         d_slow, slow_cap, sorted_from_bit, d_flag);
  hs_check_runs_slow_kernel<<<64, 256, 0, s>>>(d_ids_sorted, d_ints, K, d_slow, slow_cap, 0,
                                               d_keys_sorted, d_flag);
  return hipGetLastError();
}

hipError_t hs_launch_dir_tuples(const uint32_t* d_dir_start, const uint32_t* d_ids_sorted,
                                const int32_t* d_ints, uint32_t nb, int K, int32_t* d_dir_tuple,
                                hipStream_t s) {
  if (!nb) return hipSuccess;
  hs_dir_tuples_kernel<<<blocks_for((uint64_t)nb * K), 256, 0, s>>>(d_dir_start, d_ids_sorted,
                                                                    d_ints, nb, K, d_dir_tuple);
  return hipGetLastError();
}

hipError_t hs_launch_set_u32(uint32_t* d_p, uint32_t v, hipStream_t s) {
  hs_set_u32_kernel<<<1, 1, 0, s>>>(d_p, v);
  return hipGetLastError();
}

hipError_t hs_launch_max_u32(const uint32_t* d_in, uint32_t n, uint32_t* d_out, hipStream_t s) {
  if (!n) return hipSuccess;
  unsigned blocks = std::min(blocks_for(n), 1024u);
  hs_max_u32_kernel<<<blocks, 256, 0, s>>>(d_in, n, d_out);
  return hipGetLastError();
}

hipError_t hs_launch_pack(const uint8_t* d_codes, uint64_t n, int k, int alphabet, uint4* d_packed,
                          uint32_t* d_bad, hipStream_t s) {
  if (!n) return hipSuccess;
  hs_pack_kernel<<<blocks_for(n), 256, 0, s>>>(d_codes, n, k, hs_packed_words(k), (uint32_t)alphabet,
                                               d_packed, d_bad);
  return hipGetLastError();
}

hipError_t hs_launch_gather_packed(const uint4* d_packed_all, const uint32_t* d_ids_sorted,
                                   uint64_t n, int PW, uint4* d_out, hipStream_t s) {
  if (!n) return hipSuccess;
  hs_gather_packed_kernel<<<blocks_for(n * (uint64_t)PW), 256, 0, s>>>(d_packed_all, d_ids_sorted, n,
                                                                       PW, d_out);
  return hipGetLastError();
}

hipError_t hs_launch_probe(const hs_tables_dev& tabs, const int32_t* d_qints, uint32_t nq, int K,
                           int L, uint32_t seed, uint32_t* d_qstart, uint32_t* d_qcount,
                           uint32_t* d_nslices, uint64_t* d_cand_out,
                           unsigned long long* d_cand_total, uint32_t* d_slow,
                           const uint32_t* d_dir_base, uint32_t nb_total, uint32_t* d_bucket_count,
                           uint32_t* d_qbucket, uint32_t* d_qrank, hipStream_t s) {
  if (!nq) return hipSuccess;
  hipError_t e = hipMemsetAsync(d_slow, 0, 4, s);
  if (e != hipSuccess) return e;
  if (d_bucket_count) {
    e = hipMemsetAsync(d_bucket_count, 0, ((size_t)nb_total + 2) * 4, s);
    if (e != hipSuccess) return e;
  }
  if (tabs.probe_list && !tabs.n_list) return hipSuccess;  // (a part without a probe in this batch)
  hs_probe_kernel<<<blocks_for(tabs.probe_list ? (uint64_t)tabs.n_list : (uint64_t)nq * L), 256, 0, s>>>(
      tabs, d_qints, nq, K, L, seed, d_qstart, d_qcount, d_nslices, d_cand_out, d_cand_total, d_slow,
      d_dir_base, nb_total, d_bucket_count, d_qbucket, d_qrank);
  hs_probe_slow_kernel<<<64, 256, 0, s>>>(tabs, d_qints, K, L, seed, d_qstart, d_qcount, d_nslices,
                                          d_cand_out, d_cand_total, d_slow, d_dir_base, nb_total,
                                          d_bucket_count, d_qbucket, d_qrank);
  return hipGetLastError();
}

// Are the centres k-mers?  The reference's usual centres are (KmerToCoordinates, hclust2.cpp:49-62): every
// group of 8 doubles is then a row of the coordinate table, bit for bit.  One thread per (query, position):
// the first residue whose row equals the group, or a count of groups that equal none.  A batch of
// recognised queries runs from the codes (hs_query_codes's path): identical results -- the same doubles
// enter the same operations -- with k bytes per query in place of 64 k wherever a row is fetched.
__global__ __launch_bounds__(256) void hs_recognise_kmers_kernel(const double* __restrict__ centers, uint64_t total,
                                                                 const double* __restrict__ coords, int alphabet,
                                                                 uint8_t* __restrict__ out_codes,
                                                                 uint32_t* __restrict__ n_unrecognised) {
  __shared__ unsigned long long s_rows[HS_ALPHABET_PAD * 8];
  for (int t = threadIdx.x; t < HS_ALPHABET_PAD * 8; t += 256)
    s_rows[t] = t < alphabet * 8 ? (unsigned long long)__double_as_longlong(coords[t]) : 0ull;
  __syncthreads();
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  unsigned long long g[8];
  const double2* src = reinterpret_cast<const double2*>(centers + t * 8);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double2 v = src[j];
    g[2 * j] = (unsigned long long)__double_as_longlong(v.x);
    g[2 * j + 1] = (unsigned long long)__double_as_longlong(v.y);
  }
  int code = -1;
  for (int c = alphabet - 1; c >= 0; --c) {
    unsigned long long diff = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) diff |= g[j] ^ s_rows[c * 8 + j];
    if (!diff) code = c;
  }
  out_codes[t] = (uint8_t)(code < 0 ? 0 : code);
  if (code < 0) atomicAdd(n_unrecognised, 1u);
}

hipError_t hs_launch_recognise_kmers(const double* d_centers, uint64_t nq, int k, const double* d_coords, int alphabet,
                                     uint8_t* d_out_codes, uint32_t* d_n_unrecognised, hipStream_t s) {
  const uint64_t total = nq * (uint64_t)k;
  if (!total) return hipSuccess;
  hs_recognise_kmers_kernel<<<blocks_for(total), 256, 0, s>>>(d_centers, total, d_coords, alphabet, d_out_codes,
                                                             d_n_unrecognised);
  return hipGetLastError();
}

hipError_t hs_launch_check_codes(const uint8_t* d_in, uint64_t n_bytes, int alphabet, uint8_t* d_out,
                                 uint32_t* d_bad, hipStream_t s) {
  if (!n_bytes) return hipSuccess;
  hs_check_codes_kernel<<<blocks_for((n_bytes + 15) / 16), 256, 0, s>>>(d_in, n_bytes, (uint32_t)alphabet, d_out, d_bad);
  return hipGetLastError();
}

hipError_t hs_launch_self_probe(const hs_tables_dev& tabs, uint32_t first_id, uint32_t nq, int L,
                                uint32_t* d_qstart, uint32_t* d_qcount, uint32_t* d_nslices,
                                uint64_t* d_cand_out, unsigned long long* d_cand_total,
                                const uint32_t* d_dir_base, uint32_t nb_total, uint32_t* d_bucket_count,
                                uint32_t* d_qbucket, uint32_t* d_qrank, hipStream_t s) {
  if (!nq) return hipSuccess;
  if (d_bucket_count) {
    hipError_t e = hipMemsetAsync(d_bucket_count, 0, ((size_t)nb_total + 2) * 4, s);
    if (e != hipSuccess) return e;
  }
  hs_self_probe_kernel<<<blocks_for((uint64_t)nq * L), 256, 0, s>>>(tabs, first_id, nq, L, d_qstart, d_qcount,
                                                                    d_nslices, d_cand_out, d_cand_total,
                                                                    d_dir_base, d_bucket_count, d_qbucket,
                                                                    d_qrank);
  return hipGetLastError();
}

hipError_t hs_launch_qtables(const double* d_centers, uint32_t nq, int k, const double* d_coords,
                             int alphabet, float* d_tq, hipStream_t s) {
  if (!nq) return hipSuccess;
  const unsigned blocks = (unsigned)std::min<uint64_t>(blocks_for((uint64_t)nq * k * HS_TROW), 256u * 16u);
  hs_qtables_kernel<<<blocks, 256, 0, s>>>(d_centers, nq, k, d_coords, alphabet, d_tq);
  return hipGetLastError();
}

hipError_t hs_launch_verify(const hs_tables_dev& tabs, const uint32_t* d_qstart,
                            const uint32_t* d_qcount, const uint32_t* d_slice_off, uint32_t nql,
                            const float* d_tq, int k, int L, float r2_hi, uint32_t* d_prov_count,
                            uint32_t prov_cap, uint2* d_prov, int n_blocks, hipStream_t s) {
  if (!nql) return hipSuccess;
  const int PW = hs_packed_words(k);
#define HS_VERIFY(P)                                                                              \
  hs_verify_kernel<P, false><<<n_blocks, 256, 0, s>>>(tabs, nullptr, 0u, d_qstart, d_qcount,      \
                                                      d_slice_off, nql, d_tq, k, L, r2_hi,         \
                                                      d_prov_count, prov_cap, d_prov, nullptr, nullptr)
  if (PW == 1) HS_VERIFY(1);
  else if (PW == 2) HS_VERIFY(2);
  else if (PW == 3) HS_VERIFY(3);
  else return hipErrorInvalidValue;
#undef HS_VERIFY
  return hipGetLastError();
}

hipError_t hs_launch_bruteforce(const uint4* d_packed_all, uint32_t n, const float* d_tq,
                                uint32_t nq, int k, float r2_hi, uint32_t* d_prov_count,
                                uint32_t prov_cap, uint2* d_prov, const float* d_q_thr,
                                float* d_slice_min, int n_blocks, hipStream_t s) {
  if (!nq || !n) return hipSuccess;
  const int PW = hs_packed_words(k);
  hs_tables_dev none = {};
#define HS_BRUTE(P)                                                                               \
  hs_verify_kernel<P, true><<<n_blocks, 256, 0, s>>>(none, d_packed_all, n, nullptr, nullptr,      \
                                                     nullptr, nq, d_tq, k, 1, r2_hi, d_prov_count, \
                                                     prov_cap, d_prov, d_q_thr, d_slice_min)
  if (PW == 1) HS_BRUTE(1);
  else if (PW == 2) HS_BRUTE(2);
  else if (PW == 3) HS_BRUTE(3);
  else return hipErrorInvalidValue;
#undef HS_BRUTE
  return hipGetLastError();
}

hipError_t hs_launch_finalize(const hs_tables_dev& tabs, const uint8_t* d_codes,
                              const double* d_centers, const uint8_t* d_qcodes, const double* d_coords,
                              const uint32_t* d_qstart, const uint32_t* d_qcount,
                              const uint2* d_prov, const uint32_t* d_prov_count, uint32_t prov_cap,
                              const uint32_t* d_sorted_ql, int k, int L, double r2, double r_sqrt,
                              uint32_t q_base, uint32_t self_first, uint32_t* d_hit_count,
                              uint32_t hit_cap, uint64_t* d_hit_key, uint64_t* d_hit_val, uint32_t* d_qcnt,
                              int alphabet, const uint4* d_qpacked, uint32_t* d_hit_rank, hipStream_t s) {
  if (d_qpacked && alphabet <= HS_FIN_TABLE_ALPHABET) {  // the queries are k-mers: terms from a table
    static bool lds_granted = false;  // (static + dynamic LDS above 64 KB: ask once per process)
    if (!lds_granted) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(hs_finalize_codes_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         HS_FIN_TABLE_ALPHABET * HS_FIN_TABLE_ALPHABET * 80);
      if (e != hipSuccess) return e;
      lds_granted = true;
    }
    hs_finalize_codes_kernel<<<512, 64 * HS_FINC_WAVES, (size_t)alphabet * alphabet * 80, s>>>(
        tabs, d_qpacked, d_coords, alphabet, d_qstart, d_qcount, d_prov, d_prov_count, prov_cap, d_sorted_ql, k, L, r2,
        r_sqrt, q_base, self_first, d_hit_count, hit_cap, d_hit_key, d_hit_val, d_qcnt, d_hit_rank);
  } else if (d_qcodes)  // ... with a large alphabet: centre rows from the coordinate table
    hs_finalize_kernel<true><<<1024, 256, 0, s>>>(tabs, d_codes, nullptr, d_qcodes, d_coords, d_qstart, d_qcount,
                                                  d_prov, d_prov_count, prov_cap, d_sorted_ql, k, L, r2,
                                                  r_sqrt, q_base, self_first, d_hit_count, hit_cap,
                                                  d_hit_key, d_hit_val, d_qcnt, d_hit_rank);
  else
    hs_finalize_kernel<false><<<1024, 256, 0, s>>>(tabs, d_codes, d_centers, nullptr, d_coords, d_qstart,
                                                   d_qcount, d_prov, d_prov_count, prov_cap, d_sorted_ql, k, L,
                                                   r2, r_sqrt, q_base, self_first, d_hit_count, hit_cap,
                                                   d_hit_key, d_hit_val, d_qcnt, d_hit_rank);
  return hipGetLastError();
}
hipError_t hs_launch_hit_order(const uint64_t* d_key, const uint64_t* d_val, const uint32_t* d_hit_count,
                               uint32_t hit_cap, uint32_t q_base, uint32_t nq, const uint32_t* d_qoff,
                               const uint32_t* d_rank, void* d_kv /* hit_cap x 16 bytes */, uint32_t* d_big,
                               uint32_t* d_qlist /* 8 + 3 nq words, the first eight zero */,
                               uint32_t* d_q, uint32_t* d_id, uint32_t* d_table, double* d_dist,
                               uint64_t out_room, int n_cu, hipStream_t s) {
  if (!nq) return hipSuccess;
  ulonglong2* const kv = reinterpret_cast<ulonglong2*>(d_kv);
  hs_hit_place_kernel<<<std::max(n_cu, 1) * 8, 256, 0, s>>>(d_key, d_val, d_hit_count, hit_cap, q_base, d_qoff,
                                                            d_rank, kv);
  hs_hit_order_kernel<<<blocks_for(nq), 256, 0, s>>>(d_qoff, nq, d_hit_count, hit_cap, kv, d_big,
                                                    d_qlist, d_q, d_id, d_table, d_dist, out_room);
  // (blocks that find their list empty leave at once: a batch of few hits pays two empty launches)
  const unsigned cu = (unsigned)std::max(n_cu, 1);
  hs_hit_order_block_kernel<1024u, 256u><<<std::min(nq, cu * 8u), 256, 0, s>>>(d_qoff, nq, d_hit_count, hit_cap, 0u,
                                                                             kv, d_qlist, d_q, d_id,
                                                                             d_table, d_dist, out_room);
  hs_hit_order_block_kernel<HS_ORDER_BLOCK_MAX, 1024u><<<std::min(nq, cu * 2u), 1024, 0, s>>>(
      d_qoff, nq, d_hit_count, hit_cap, 1u, kv, d_qlist, d_q, d_id, d_table, d_dist, out_room);
  // (the unbucketed keys are its scratch: nothing reads them again unless *d_big, and then it does not run)
  hs_hit_order_huge_kernel<<<std::min(nq, cu), 1024, 0, s>>>(d_qoff, nq, d_hit_count, hit_cap, d_big, kv,
                                                            d_qlist, const_cast<uint64_t*>(d_key), d_q, d_id, d_table,
                                                            d_dist, out_room);
  return hipGetLastError();
}

hipError_t hs_launch_windows(const uint8_t* d_residues, uint32_t n_residues, const uint32_t* d_seq_start,
                             const uint32_t* d_win_off, uint32_t n_seq, int k, uint8_t* d_codes,
                             uint32_t* d_win_pos, hipStream_t s) {
  if (!n_residues || !n_seq) return hipSuccess;
  hs_windows_kernel<<<blocks_for(n_residues), 256, 0, s>>>(d_residues, n_residues, d_seq_start, d_win_off,
                                                          n_seq, k, d_codes, d_win_pos);
  return hipGetLastError();
}

hipError_t hs_launch_klsh(const uint8_t* d_classes, const uint64_t* d_seq_start, uint64_t n_seq,
                          const double* d_w, const double* d_b, const double* d_t, uint32_t bits,
                          uint64_t* d_codes, uint64_t* d_uncertain, hipStream_t s) {
  if (!n_seq) return hipSuccess;
  const unsigned blocks = (unsigned)std::min<uint64_t>((n_seq + 3) / 4, 1u << 16);
  hs_klsh_kernel<<<blocks, 256, 0, s>>>(d_classes, d_seq_start, n_seq, d_w, d_b, d_t, bits, d_codes,
                                        d_uncertain);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hs_giant_buckets_kernel(const int32_t* __restrict__ tuple, int K,
                                                               const uint32_t* __restrict__ start, uint32_t nb,
                                                               uint32_t threshold, uint64_t* __restrict__ out,
                                                               uint32_t cap, uint32_t* __restrict__ count) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b >= nb || start[b + 1] - start[b] <= threshold) return;
  const uint32_t at = atomicAdd(count, 1u);
  if (at < cap) out[at] = hs_tuple_hash(tuple + (uint64_t)b * K, K, 1);
}
hipError_t hs_launch_giant_buckets(const int32_t* d_dir_tuple, int K, const uint32_t* d_dir_start, uint32_t nb,
                                   uint32_t threshold, uint64_t* d_out, uint32_t cap, uint32_t* d_count, hipStream_t s) {
  if (!nb) return hipSuccess;
  hs_giant_buckets_kernel<<<blocks_for(nb), 256, 0, s>>>(d_dir_tuple, K, d_dir_start, nb, threshold, d_out, cap, d_count);
  return hipGetLastError();
}

// Bucket partition, first pass over a batch's probes: whose is it?  From the bucket ints alone (hs_tuple_hash:
// ~ 1/10 of the fingerprint's instructions); a probe of another part gets the outputs of a probe that found
// nothing.  The part's own probes then go through hs_probe_kernel as a list (1 / n_parts of the batch).
__global__ __launch_bounds__(256) void hs_part_owned_kernel(hs_tables_dev tabs, const int32_t* __restrict__ qints,
                                                            uint32_t nq, int K, int L, uint32_t nb_total,
                                                            uint32_t* __restrict__ flag, uint32_t* __restrict__ qstart,
                                                            uint32_t* __restrict__ qcount, uint32_t* __restrict__ nslices,
                                                            uint64_t* __restrict__ cand_out,
                                                            uint32_t* __restrict__ qbucket) {
  const uint32_t ql = blockIdx.x * 256 + threadIdx.x;
  const uint32_t nql = nq * (uint32_t)L;
  if (ql > nql) return;
  if (ql == nql) {
    flag[ql] = 0;
    return;
  }
  const int l = (int)(ql % (uint32_t)L);
  const int32_t* tg = qints + (uint64_t)ql * K;
  uint64_t th;
  if ((K & 3) == 0) {  // 16-byte loads, all in flight at once
    int4 v[HS_MAX_K / 4];
#pragma unroll
    for (int j4 = 0; j4 < HS_MAX_K / 4; ++j4)
      if (4 * j4 < K) v[j4] = reinterpret_cast<const int4*>(tg)[j4];
    int32_t t[HS_MAX_K];
#pragma unroll
    for (int j4 = 0; j4 < HS_MAX_K / 4; ++j4)
      if (4 * j4 < K) {
        t[4 * j4] = v[j4].x;
        t[4 * j4 + 1] = v[j4].y;
        t[4 * j4 + 2] = v[j4].z;
        t[4 * j4 + 3] = v[j4].w;
      }
    uint64_t h = 0x243f6a8885a308d3ull;
#pragma unroll
    for (int j = 0; j < HS_MAX_K; ++j)
      if (j < K) {
        h = (h ^ (uint32_t)t[j]) * 0x9e3779b97f4a7c15ull;
        h ^= h >> 29;
      }
    th = h;
  } else {
    th = hs_tuple_hash(tg, K, 1);
  }
  const hs_table_dev& tb = tabs.t[l];
  uint32_t glo = 0, ghi = tb.n_giant;
  while (glo < ghi) {
    const uint32_t mid = (glo + ghi) >> 1;
    if (tb.giant_key[mid] < th) glo = mid + 1; else ghi = mid;
  }
  const bool giant = glo < tb.n_giant && tb.giant_key[glo] == th;
  const bool mine = hs_probe_part(th, giant, tabs.q_first + ql / (uint32_t)L, tabs.n_parts) == tabs.part;
  flag[ql] = mine ? 1u : 0u;
  if (!mine) {
    qstart[ql] = 0;
    qcount[ql] = 0;
    nslices[ql] = 0;
    if (cand_out) cand_out[ql] = 0;
    if (qbucket) qbucket[ql] = nb_total;
  }
}
hipError_t hs_launch_part_owned(const hs_tables_dev& tabs, const int32_t* d_qints, uint32_t nq, int K, int L,
                                uint32_t nb_total, uint32_t* d_flag, uint32_t* d_qstart, uint32_t* d_qcount,
                                uint32_t* d_nslices, uint64_t* d_cand_out, uint32_t* d_qbucket, hipStream_t s) {
  if (!nq) return hipSuccess;
  hs_part_owned_kernel<<<blocks_for((uint64_t)nq * L + 1), 256, 0, s>>>(tabs, d_qints, nq, K, L, nb_total, d_flag,
                                                                          d_qstart, d_qcount, d_nslices, d_cand_out,
                                                                          d_qbucket);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void hs_flagged_list_kernel(const uint32_t* __restrict__ flag,
                                                              const uint32_t* __restrict__ pos, uint32_t n,
                                                              uint32_t* __restrict__ list) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n && flag[i]) list[pos[i]] = i;
}
hipError_t hs_launch_flagged_list(const uint32_t* d_flag, const uint32_t* d_pos, uint32_t n, uint32_t* d_list,
                                  hipStream_t s) {
  if (!n) return hipSuccess;
  hs_flagged_list_kernel<<<blocks_for(n), 256, 0, s>>>(d_flag, d_pos, n, d_list);
  return hipGetLastError();
}

hipError_t hs_launch_dir_records(const uint64_t* d_dir_key, const uint32_t* d_dir_start, const int32_t* d_dir_tuple,
                                 uint32_t nb, int K, uint4* d_rec, uint32_t* d_flag, hipStream_t s) {
  if (!nb) return hipSuccess;
  hs_dir_records_kernel<<<blocks_for(nb), 256, 0, s>>>(d_dir_key, d_dir_start, d_dir_tuple, nb, K, d_rec, d_flag);
  return hipGetLastError();
}

hipError_t hs_launch_dir_jump(const uint64_t* d_dir_key, uint32_t nb, uint32_t shift, uint32_t n_slots,
                              uint32_t* d_jump, hipStream_t s) {
  hs_dir_jump_kernel<<<blocks_for((uint64_t)nb + 1), 256, 0, s>>>(d_dir_key, nb, shift, n_slots, d_jump);
  return hipGetLastError();
}

hipError_t hs_launch_invert_perm(const uint32_t* d_perm, uint32_t n, uint32_t* d_out, hipStream_t s) {
  if (!n) return hipSuccess;
  hs_invert_perm_kernel<<<blocks_for(n), 256, 0, s>>>(d_perm, n, d_out);
  return hipGetLastError();
}

// out[i][0..k) = all[subset[i]][0..k)  (subset == null: identity); one thread per byte
__global__ __launch_bounds__(256) void hs_gather_rows_kernel(const uint8_t* __restrict__ all,
                                                             const uint32_t* __restrict__ subset, uint64_t n_sub,
                                                             int k, uint8_t* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_sub * (uint64_t)k) return;
  const uint64_t i = t / (uint64_t)k;
  const int p = (int)(t - i * (uint64_t)k);
  out[t] = all[(uint64_t)(subset ? subset[i] : (uint32_t)i) * k + p];
}
hipError_t hs_launch_gather_rows(const uint8_t* d_all, const uint32_t* d_subset, uint64_t n_sub, int k,
                                 uint8_t* d_out, hipStream_t s) {
  if (!n_sub) return hipSuccess;
  hs_gather_rows_kernel<<<blocks_for(n_sub * (uint64_t)k), 256, 0, s>>>(d_all, d_subset, n_sub, k, d_out);
  return hipGetLastError();
}

hipError_t hs_launch_validate_table(const uint32_t* d_ids, uint32_t n, uint32_t* d_pos_of,
                                    const uint32_t* d_dir_start, const uint64_t* d_dir_key,
                                    const int32_t* d_dir_tuple, uint32_t nb, int K, uint32_t seed,
                                    uint32_t* d_flag, uint32_t* d_max_bucket, hipStream_t s) {
  hipError_t e = hipMemsetAsync(d_pos_of, 0xff, (size_t)n * 4, s);
  if (e != hipSuccess) return e;
  if (n) hs_validate_ids_kernel<<<blocks_for(n), 256, 0, s>>>(d_ids, n, d_pos_of, d_flag);
  hs_validate_dir_kernel<<<std::max(1u, blocks_for(nb)), 256, 0, s>>>(d_dir_start, d_dir_key, d_dir_tuple,
                                                                      d_ids, nb, n, K, seed, d_flag,
                                                                      d_max_bucket);
  if (n > 1) hs_validate_order_kernel<<<blocks_for(n - 1), 256, 0, s>>>(d_ids, d_dir_start, nb, n, d_flag);
  return hipGetLastError();
}

hipError_t hs_launch_bf_finalize(const uint8_t* d_codes, const double* d_centers,
                                 const double* d_coords, const uint2* d_prov,
                                 const uint32_t* d_prov_count, uint32_t prov_cap, int k, double R,
                                 uint32_t q_base, uint32_t* d_hit_count, uint32_t hit_cap,
                                 uint64_t* d_hit_key, uint64_t* d_hit_val, hipStream_t s) {
  hs_bf_finalize_kernel<<<1024, 256, 0, s>>>(d_codes, d_centers, d_coords, d_prov, d_prov_count,
                                             prov_cap, k, R, q_base, d_hit_count, hit_cap, d_hit_key,
                                             d_hit_val);
  return hipGetLastError();
}

hipError_t hs_launch_unpack_hits(const uint64_t* d_key, const uint64_t* d_val, uint32_t n,
                                 uint32_t* d_q, uint32_t* d_id, uint32_t* d_table, double* d_dist,
                                 hipStream_t s) {
  if (!n) return hipSuccess;
  hs_unpack_hits_kernel<<<blocks_for(n), 256, 0, s>>>(d_key, d_val, n, d_q, d_id, d_table, d_dist);
  return hipGetLastError();
}

hipError_t hs_launch_kth_min(const float* d_slice_min, uint32_t nq, uint32_t per_q, uint32_t topk,
                             float* d_thr, hipStream_t s) {
  if (!nq) return hipSuccess;
  hs_kth_min_kernel<<<nq, 256, 0, s>>>(d_slice_min, per_q, topk, d_thr);
  return hipGetLastError();
}

hipError_t hs_launch_topk_exact(const uint8_t* d_codes, const double* d_centers,
                                const double* d_coords, const uint2* d_prov,
                                const uint32_t* d_prov_count, uint32_t prov_cap, int k,
                                uint64_t* d_key, uint64_t* d_val, hipStream_t s) {
  hs_topk_exact_kernel<<<1024, 256, 0, s>>>(d_codes, d_centers, d_coords, d_prov, d_prov_count,
                                            prov_cap, k, d_key, d_val);
  return hipGetLastError();
}
