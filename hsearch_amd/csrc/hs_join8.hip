// hs_join8.hip -- int8 variant of the bucket-join filter (see hs_join.hip for the structure).
//
// Same work items, same LDS staging and survivor protocol as hs_join_kernel, but the lower-bound
// filter is evaluated in FIXED POINT on v_mfma_i32_32x32x32_i8 (K = 32 per instruction at the
// cycles of the fp16 K = 16 form): 4 MFMAs per 32x32 tile instead of 7.
//
// Quantisation.  s = 127 / max |coordinate| over the 4 table columns the filter uses;
// x^ = rint(s x), c^ = rint(s c) saturated to +-127 (int8; the extra error of a saturated query
// coordinate is charged to gamma, and a query far outside the table's range makes the batch fall
// back to the fp16 join).  With x = (x^ + e)/s, c = (c^ + n)/s, |e|, |n| <= 1/2:
//     s^2 x.c = x^.c^ + x^.n + e.c^ + e.n,   |x^.n| <= L1(x^)/2, |e.c^| <= L1(c^)/2, |e.n| <= dims/4.
// A pair with exact d2 <= R^2 has |x1 - c1|^2 <= R^2, i.e. x1.c1 >= (|x1|^2 + |c1|^2 - R^2)/2, hence
//     x^.c^  >=  rho(x) + gamma(c),
//     rho   = floor(s^2 |x1|^2 / 2 - L1(x^)/2 - dims/4 - 2),
//     gamma = floor(s^2 (|c1|^2 - R^2) / 2 - L1(c^)/2 - 2)             (the 2s absorb fp rounding).
// The filter passes a pair iff acc = x^.c^ - rho - gamma >= 0, all in exact int32 arithmetic.
//
// K layout (128 = 4 k-steps x 2 lane halves x 16 bytes): byte 4p + j = coordinate j of position p
// for p < 25; the 28 bytes of positions 25..31 are spare and carry -rho and -gamma as products of
// base-127 digits: slots 0..13: A = digits of rho, B = (-127 x13, -1); slots 14..27: A = (127 x13,
// 1), B = digits of -gamma.  A value too large for 13 digits is clamped in the permissive direction
// (the filter may only pass MORE); a gamma too negative to represent marks the batch unsafe.
#include <stdlib.h>

#include <algorithm>

#include "hs_internal.h"

namespace {

typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx16 __attribute__((ext_vector_type(16)));

constexpr int QD = 4;        // table columns used
constexpr int QROW = 128;    // bytes of a quantised query row (global)
constexpr int QPIECES = 8;   // 16-byte pieces per row
constexpr int QLROW = 144;   // LDS row stride in bytes (9 x 16: conflict-free b128)
constexpr int JQ = 32;       // queries per MFMA column tile
constexpr int JC = 64;       // queries per LDS chunk (one barrier): two column tiles
constexpr int JT = 4;        // 32-member row tiles per wave
constexpr int JM = 4 * JT * 32;  // must equal hs_join.hip's JM (work items are shared)
constexpr uint32_t JRES = 64;
constexpr int DIG = 13;      // base-127 digits (+1 remainder slot) per threshold term
constexpr int DIGMAX = 127 * DIG;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// 14 signed int8 slots d[0..13] with 127 * (d0 + ... + d12) + d13 == v, or the closest value on
// the `permissive` side when v does not fit (lower for rho: the filter only gets more permissive).
__device__ __forceinline__ void digits127(int v, int (&d)[DIG + 1], bool* overflow_high) {
  // floor division by 127 for either sign
  int q = v / 127;
  int rem = v - q * 127;
  if (rem < 0) {
    rem += 127;
    q -= 1;
  }
  if (q > DIGMAX) {  // too large: clamp down
    q = DIGMAX;
    rem = 126;
    if (overflow_high) *overflow_high = true;
  }
  if (q < -DIGMAX) {  // too small: clamp up is only allowed for the caller that says so
    q = -DIGMAX;
    rem = 0;
  }
#pragma unroll
  for (int j = 0; j < DIG; ++j) {
    const int take = max(-127, min(127, q));
    d[j] = take;
    q -= take;
  }
  d[DIG] = rem;
}

// ---------------------------------------------------------------------------------- tables
// tab8[aa] = { packed x^ (4 int8), |x1|^2 as float bits, L1(x^), 0 }; scale[0] = s, scale[1] = s^2/2
__global__ void hs_jtables8_kernel(const double* __restrict__ coords, int alphabet,
                                   uint4* __restrict__ tab8, float* __restrict__ scale,
                                   uint32_t* __restrict__ unsafe) {
  __shared__ double smax[32];
  const int aa = threadIdx.x;
  if (aa >= 32) return;
  double m = 0.0;
  for (int j = 0; j < QD; ++j) m = fmax(m, aa < alphabet ? fabs(coords[aa * 8 + j]) : 0.0);
  smax[aa] = m;
  __syncthreads();
  double mm = 0.0;
  for (int i = 0; i < 32; ++i) mm = fmax(mm, smax[i]);
  if (!(mm > 0.0) || !(mm < 1e6)) {
    if (aa == 0) atomicOr(unsafe, 1u);
    mm = 1.0;
  }
  const double s = 127.0 / mm;
  uint32_t pack = 0;
  int l1 = 0;
  double n = 0.0;
  for (int j = 0; j < QD; ++j) {
    const double v = aa < alphabet ? coords[aa * 8 + j] : 0.0;
    int q = (int)rint(s * v);
    q = max(-127, min(127, q));
    pack |= ((uint32_t)(q & 0xff)) << (8 * j);
    l1 += abs(q);
    n += v * v;
  }
  tab8[aa] = make_uint4(pack, __float_as_uint((float)n), (uint32_t)l1, 0u);
  if (aa == 0) {
    scale[0] = (float)s;
    scale[1] = (float)(0.5 * s * s);
  }
}

// ---------------------------------------------------------------------------------- query prep
// c8[q] (128 bytes): byte 4p + j = c^ of coordinate j (< 4) of position p (< min(k, 25)), zeros
// up to byte 99; spare slots at bytes 100..127: (-127 x13, -1) then the digits of -gamma.
__global__ __launch_bounds__(256) void hs_qprep8_kernel(const double* __restrict__ centers, uint32_t nq,
                                                        int k, double r2, const float* __restrict__ scale,
                                                        int8_t* __restrict__ c8,
                                                        uint32_t* __restrict__ unsafe) {
  const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;
  const int lane = lane_id();
  const double s = (double)scale[0];
  const double* c = centers + (uint64_t)q * 8 * k;
  int8_t* out = c8 + (uint64_t)q * QROW;
  double nc = 0.0, pen = 0.0;
  int l1 = 0;
  bool bad = false;
  for (int i = lane; i < 100; i += 64) {
    const int pos = i >> 2, j = i & 3;
    int qv = 0;
    if (pos < k) {
      const double v = c[8 * pos + j];
      nc += v * v;
      const double sv = s * v;
      bad = bad || !(fabs(sv) < 1.0e6);
      // saturate; a coordinate outside +-127 has a quantisation error n_i > 1/2, which costs at
      // most (127 + 1/2)(|n_i| - 1/2) more in the bound (|x^_i| <= 127, |e_i| <= 1/2)
      qv = (int)fmax(-127.0, fmin(127.0, rint(sv)));
      pen += 127.5 * fmax(0.0, fabs(sv - (double)qv) - 0.5);
      l1 += abs(qv);
    }
    out[i] = (int8_t)qv;
  }
  for (int off = 32; off; off >>= 1) {
    nc += __shfl_xor(nc, off);
    pen += __shfl_xor(pen, off);
    l1 += __shfl_xor(l1, off);
  }
  // gamma = floor(s^2 (nc - R^2)/2 - L1/2 - saturation penalty - 2)
  const double g = floor(0.5 * s * s * (nc - r2) - 0.5 * (double)l1 - pen - 2.0);
  // a query far outside the table's range would make the filter uselessly permissive
  bad = bad || !(fabs(g) < 1.0e9) || !(pen < 30000.0);
  int d[DIG + 1];
  const int v = bad ? 0 : -(int)g;
  // -gamma too LARGE for the digits would have to be clamped in the non-permissive direction
  bool too_high = false;
  digits127(v, d, &too_high);
  bad = bad || too_high;
  if (__ballot(bad) && lane == 0) atomicOr(unsafe, 1u);
  if (lane < 28) {
    int8_t b;
    if (lane < DIG) b = (int8_t)-127;
    else if (lane == DIG) b = (int8_t)-1;
    else b = (int8_t)d[lane - (DIG + 1)];
    out[100 + lane] = b;
  }
}

// rows in SEGMENT order: a chunk of 32 probing queries is one contiguous 4 KB block
__global__ __launch_bounds__(256) void hs_gather_c8_kernel(const int8_t* __restrict__ c8,
                                                           const uint32_t* __restrict__ sorted_ql,
                                                           uint32_t nql, int L,
                                                           int8_t* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (uint64_t)nql * QPIECES) return;
  const uint32_t p = (uint32_t)(t / QPIECES);
  const int g = (int)(t - (uint64_t)p * QPIECES);
  const uint32_t q = sorted_ql[p] / (uint32_t)L;
  *reinterpret_cast<uint4*>(out + (uint64_t)p * QROW + g * 16) =
      *reinterpret_cast<const uint4*>(c8 + (uint64_t)q * QROW + g * 16);
}

// ------------------------------------------------------------------------------------------ join
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

__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) {
  return ((uint32_t)a & 0xffu) | (((uint32_t)b & 0xffu) << 8) | (((uint32_t)c & 0xffu) << 16) |
         (((uint32_t)d & 0xffu) << 24);
}

// A operands (4 k-steps) of one 32-member row tile for lane (r, h): k-step s < 3 carries positions
// 8s + 4h + {0,1,2,3}; k-step 3 carries position 24 and the spare slots.
__device__ __forceinline__ void build_afrags8(const uint4 pk, int h, int k, const uint4* sTab8,
                                              float s2half, intx4 (&A)[4]) {
  // lanes of the upper half take positions 4..7, 12..15, ...: shift the word down by 20 bits
  const uint32_t sh = 20u * (uint32_t)h;
  const uint32_t x = __funnelshift_r(pk.x, pk.y, sh), y = __funnelshift_r(pk.y, pk.z, sh),
                 z = __funnelshift_r(pk.z, pk.w, sh), w = pk.w >> sh;
  float nx = 0.f;
  int l1 = 0;
#define HS_A8(S, M)                                                                  \
  {                                                                                  \
    const uint4 row = sTab8[residue_at<40 * S + 5 * M>(x, y, z, w)];                 \
    A[S][M] = (int)row.x;                                                            \
    const bool real = 8 * S + 4 * h + M < k;                                         \
    nx += real ? __uint_as_float(row.y) : 0.f;                                       \
    l1 += real ? (int)row.z : 0;                                                     \
  }
  HS_A8(0, 0) HS_A8(0, 1) HS_A8(0, 2) HS_A8(0, 3)
  HS_A8(1, 0) HS_A8(1, 1) HS_A8(1, 2) HS_A8(1, 3)
  HS_A8(2, 0) HS_A8(2, 1) HS_A8(2, 2) HS_A8(2, 3)
#undef HS_A8
  const uint4 row24 = sTab8[(pk.w >> 24) & 31u];  // position 24 sits at bit 120 of the unshifted word
  if (h == 0 && 24 < k) {
    nx += __uint_as_float(row24.y);
    l1 += (int)row24.z;
  }
  nx += __shfl_xor(nx, 32);
  l1 += __shfl_xor(l1, 32);
  const int dims = QD * min(k, 25);
  const int rho = (int)floorf(s2half * nx - 0.5f * (float)l1 - 0.25f * (float)dims - 2.0f);
  int d[DIG + 1];
  digits127(rho, d, nullptr);  // too large -> clamped down: more permissive, never less
  if (h == 0) {
    A[3][0] = 24 < k ? (int)row24.x : 0;
    A[3][1] = (int)pack4(d[0], d[1], d[2], d[3]);
    A[3][2] = (int)pack4(d[4], d[5], d[6], d[7]);
    A[3][3] = (int)pack4(d[8], d[9], d[10], d[11]);
  } else {
    A[3][0] = (int)pack4(d[12], d[13], 127, 127);
    A[3][1] = (int)pack4(127, 127, 127, 127);
    A[3][2] = (int)pack4(127, 127, 127, 127);
    A[3][3] = (int)pack4(127, 127, 127, 1);
  }
}

__global__ __launch_bounds__(256, 2) void hs_join8_kernel(
    const uint4* __restrict__ desc, uint32_t n_items, const uint4* __restrict__ packed_base,
    const uint32_t* __restrict__ sorted_ql, const int8_t* __restrict__ c8s,
    const uint4* __restrict__ tab8, const float* __restrict__ scale, int k,
    uint32_t* __restrict__ prov_count, uint32_t prov_cap, uint2* __restrict__ prov) {
  __shared__ __attribute__((aligned(16))) int8_t sB[2][JC * QLROW];
  __shared__ uint4 sTab8[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  if (tid < 32) sTab8[tid] = tab8[tid];
  const float s2half = scale[1];
  __syncthreads();
  // two 16-byte pieces of a chunk per thread: rows tid / 8 and 32 + tid / 8, piece tid % 8
  const int dst = (tid >> 3) * QLROW + (tid & 7) * 16;
  const int boff = r * QLROW + h * 16;  // B operand of k-step s: boff + 32 s (bytes)
  int buf = 0;
  uint32_t item = blockIdx.x;
  if (item >= n_items) return;
  uint32_t res_base = 0, res_used = JRES;
  if (blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_sleep(8);
  uint4 d0 = desc[2 * (uint64_t)item], d1 = desc[2 * (uint64_t)item + 1];
  uint4 pk[JT], pre, pre2;
  {
    const uint4* packed = packed_base + (int64_t)(((uint64_t)d0.y << 32) | (uint64_t)d0.x);
    const uint32_t idx = d0.w * JM + wave * (32 * JT) + r;
#pragma unroll
    for (int t = 0; t < JT; ++t) pk[t] = packed[min(idx + 32 * t, d0.z - 1)];
    const uint4* src = reinterpret_cast<const uint4*>(c8s + (uint64_t)(d1.x + d1.y) * QROW);
    pre = src[tid];
    pre2 = src[tid + 256];
  }
  while (true) {
    const uint32_t M = d0.z, mt = d0.w;
    const uint32_t qoff = d1.x, q_begin = d1.y, q_end = d1.z, mstart = d1.w;
    const uint32_t wbase = mt * JM + wave * (32 * JT);
    const bool wave_on = wbase < M;
    const uint32_t next_item = item + gridDim.x;
    const bool has_next = next_item < n_items;
    uint4 nd0 = d0, nd1 = d1;
    if (has_next) {
      nd0 = desc[2 * (uint64_t)next_item];
      nd1 = desc[2 * (uint64_t)next_item + 1];
    }
    intx4 A[JT][4];
    if (wave_on) {
#pragma unroll
      for (int t = 0; t < JT; ++t) build_afrags8(pk[t], h, k, sTab8, s2half, A[t]);
    }
    for (uint32_t qc0 = q_begin; qc0 < q_end; qc0 += JC) {
      int8_t* tile0 = sB[buf];
      *reinterpret_cast<uint4*>(&tile0[dst]) = pre;
      *reinterpret_cast<uint4*>(&tile0[dst + JQ * QLROW]) = pre2;
      __syncthreads();
      {
        const bool more = qc0 + JC < q_end;
        const uint64_t row = more ? (uint64_t)(qoff + qc0 + JC) : (uint64_t)(nd1.x + nd1.y);
        const uint4* src = reinterpret_cast<const uint4*>(c8s + row * QROW);
        pre = src[tid];
        pre2 = src[tid + 256];
      }
      if (qc0 == q_begin) {
        const uint4* packed = packed_base + (int64_t)(((uint64_t)nd0.y << 32) | (uint64_t)nd0.x);
        const uint32_t idx = nd0.w * JM + wave * (32 * JT) + r;
#pragma unroll
        for (int t = 0; t < JT; ++t) pk[t] = packed[min(idx + 32 * t, nd0.z - 1)];
      }
      buf ^= 1;
      if (!wave_on) continue;
#pragma unroll 1
      for (int half = 0; half < JC / JQ; ++half) {
      const uint32_t qc = qc0 + (uint32_t)(half * JQ);
      if (qc >= q_end) break;  // uniform: the second column tile of a ragged last chunk
      const int8_t* tile = tile0 + half * JQ * QLROW;
      intx16 acc[JT];
#pragma unroll
      for (int t = 0; t < JT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0;
      intx4 b0 = *reinterpret_cast<const intx4*>(&tile[boff]);
      intx4 b1 = *reinterpret_cast<const intx4*>(&tile[boff + 32]);
#define HS_STEP8(S, B)                                                                       \
  _Pragma("unroll") for (int t = 0; t < JT; ++t)                                             \
      acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[t][S], B, acc[t], 0, 0, 0);           \
  if (S + 2 < 4) B = *reinterpret_cast<const intx4*>(&tile[boff + 32 * (S + 2)]);
      HS_STEP8(0, b0) HS_STEP8(1, b1) HS_STEP8(2, b0) HS_STEP8(3, b1)
#undef HS_STEP8
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, JT, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * JT, 0);
      // ---- survivors: acc >= 0 (sign bit clear).  "all negative" = sign bit of the AND of all
      //      accumulators.  D layout: col = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 h.
      uint32_t sall = 0xffffffffu;
#pragma unroll
      for (int t = 0; t < JT; ++t) {
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o[j] = (uint32_t)acc[t][4 * j] & (uint32_t)acc[t][4 * j + 1] & (uint32_t)acc[t][4 * j + 2] &
                 (uint32_t)acc[t][4 * j + 3];
        sall &= (o[0] & o[1]) & (o[2] & o[3]);
      }
      if (__ballot((int)sall >= 0)) {
        const bool col_ok = qc + (uint32_t)r < q_end;
        const uint32_t ql = col_ok ? sorted_ql[qoff + qc + r] : 0u;
#pragma unroll
        for (int t = 0; t < JT; ++t) {
          uint32_t mask = 0;
#pragma unroll
          for (int i = 0; i < 16; ++i) mask |= ((~(uint32_t)acc[t][i]) >> 31) << i;
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
              const uint32_t cnt = (uint32_t)__popcll(m);
              if (res_used + cnt > JRES) {
                if (res_used < JRES && lane >= (int)res_used && res_base + lane < prov_cap)
                  prov[res_base + lane] = make_uint2(0xffffffffu, 0u);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(prov_count, (uint32_t)JRES);
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

hipError_t hs_launch_jtables8(const double* d_coords, int alphabet, void* d_tab8, float* d_scale,
                              uint32_t* d_unsafe, hipStream_t s) {
  hs_jtables8_kernel<<<1, 32, 0, s>>>(d_coords, alphabet, (uint4*)d_tab8, d_scale, d_unsafe);
  return hipGetLastError();
}

hipError_t hs_launch_qprep8(const double* d_centers, uint32_t nq, int k, double r2,
                            const float* d_scale, void* d_c8, uint32_t* d_unsafe, hipStream_t s) {
  if (!nq) return hipSuccess;
  hs_qprep8_kernel<<<blocks_for(nq, 4), 256, 0, s>>>(d_centers, nq, k, r2, d_scale, (int8_t*)d_c8,
                                                     d_unsafe);
  return hipGetLastError();
}

hipError_t hs_launch_gather_c8(const void* d_c8, const uint32_t* d_sorted_ql, uint32_t nql, int L,
                               void* d_out, hipStream_t s) {
  if (!nql) return hipSuccess;
  hs_gather_c8_kernel<<<blocks_for((uint64_t)nql * QPIECES), 256, 0, s>>>(
      (const int8_t*)d_c8, d_sorted_ql, nql, L, (int8_t*)d_out);
  return hipGetLastError();
}

hipError_t hs_launch_join8(const uint4* d_desc, uint32_t n_items, const uint4* d_packed_base,
                           const uint32_t* d_sorted_ql, const void* d_c8s, const void* d_tab8,
                           const float* d_scale, int k, uint32_t* d_prov_count, uint32_t prov_cap,
                           uint2* d_prov, int n_blocks, hipStream_t s) {
  if (!n_items) return hipSuccess;
  hs_join8_kernel<<<n_blocks, 256, 0, s>>>(d_desc, n_items, d_packed_base, d_sorted_ql,
                                           (const int8_t*)d_c8s, (const uint4*)d_tab8, d_scale, k,
                                           d_prov_count, prov_cap, d_prov);
  return hipGetLastError();
}
