// hs_join8.hip -- int8 form of the bucket-join filter: the default verify path (hs_join.hip holds
// the segment/work-item machinery and the fp16 form it falls back to).
//
// The lower-bound filter is evaluated in FIXED POINT on v_mfma_i32_32x32x32_i8 (K = 32 per
// instruction at the cycles of the fp16 K = 16 form): 4 MFMAs per 32x32 tile instead of 7.  The
// kernel is wave-independent: no LDS staging, no workgroup barrier (see hs_join8w_kernel).
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
// base-127 digits: bytes 100..111: A = digits of rho (11 + remainder), B = (-127 x11, -1); bytes
// 112..113 unused; bytes 114..127: A = (127 x13, 1), B = digits of -gamma.  A value too large for
// its digits is clamped in the permissive direction (the filter may only pass MORE); a gamma too
// negative to represent marks the batch unsafe.
//
// Longer k-mers (26..50 residues, two packed words): the same row with more k-steps -- KS = 6 (k <= 41)
// or 8 (k <= 50) instead of 4; coordinates at bytes 4p + j as before, the 28 digit slots are always
// the LAST 28 bytes of the row (bytes 32 KS - 28 ..), the member's record is the 16 bytes
// 32 KS - 32 .. 32 KS - 17 of its row (x^ of position 8 KS - 8 if the k-mer has one, then the rho
// slots).  Such work items are 64 members (two row tiles) per wave: the A operands of 128 members
// over 6 or 8 k-steps would not leave room for the query tiles in flight.
//
// Short k-mers (k <= 20, "wide" rows): R^2 is no longer far below the typical 4-coordinate distance of
// bucket mates (k = 15: the 4-column bound passes 4 % of random pairs, the 8-column bound 0.03 %), so
// the row carries ALL 8 coordinates, quantised with one scale: byte 8p + j = coordinate j of
// position p, 6 k-steps (160 coordinate bytes + 4 spare + 28 digit slots), rho and gamma over all
// 8 columns with dims = 8k, 64 members per work item like the other 6-k-step rows; there is nothing
// left for hs_refine8_kernel to add, so it does not run.  The same rows with 8 k-steps (k = 21..25)
// serve calls whose radius is large for the k-mer length (hs_capi.hip want_wide): their member
// records are built the first time such a radius is asked for.
//
// rho depends on the member only, so it is evaluated ONCE, at index build: hs_gather_rec8_kernel
// writes, next to the bucket-ordered packed copy, a 16-byte record per entry = bytes 96..111 of the
// member's A row (x^ of position 24, then the 12 rho slots).  The join kernel then builds a member's
// A operand with 12 dword table lookups per lane and nothing else.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "hs_internal.h"

namespace {

typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx16 __attribute__((ext_vector_type(16)));

constexpr int QD = 4;        // table columns used
constexpr int HS_J8_CONST_AT = 48;  // uint4 index in the int8 table block of the gamma slots' constant factors
// k-steps of 32 bytes in a row: 4 (k <= 25), 6 (k <= 41), 8 (k <= 50); row = 32 KS bytes = 2 KS pieces
// wide rows (all 8 coordinates): 6 (k <= 20) or 8 (k <= 25)
__host__ __device__ constexpr int ks_of(int k, bool wide = false) {
  return wide ? (k <= 20 ? 6 : 8) : k <= 25 ? 4 : k <= 41 ? 6 : 8;
}
// survivor slots a wave reserves per counter access: same-address atomics complete at ~ 90 per
// microsecond, and a hit-heavy launch (k = 15 at the C2 sizes: 1.4e8 survivors in 25 ms) asked for
// 2.2e6 blocks of 64 -- the counter's whole capacity; 256 leaves it at a quarter
constexpr uint32_t JRES = 256;
// the unused tail of a wave's reserved block, marked so that the consumers skip it
__device__ __forceinline__ void close_reservation(uint2* __restrict__ prov, uint32_t res_base, uint32_t res_used,
                                                  uint32_t prov_cap, int lane) {
  for (uint32_t i = res_used + (uint32_t)lane; i < JRES; i += 64u)
    if (res_base + i < prov_cap) prov[res_base + i] = make_uint2(0xffffffffu, 0u);
}
constexpr int DIG = 13;      // base-127 digits (+1 remainder slot) of -gamma
constexpr int RDIG = 11;     // base-127 digits (+1 remainder slot) of rho

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ND + 1 signed int8 slots with 127 * (d[0] + ... + d[ND-1]) + d[ND] == v, or the closest value on
// the `permissive` side when v does not fit (lower for rho: the filter only gets more permissive).
template <int ND>
__device__ __forceinline__ void digits127(int v, int (&d)[ND + 1], bool* overflow_high) {
  constexpr int DIGMAX = 127 * ND;
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
  for (int j = 0; j < ND; ++j) {
    const int take = max(-127, min(127, q));
    d[j] = take;
    q -= take;
  }
  d[ND] = rem;
}

// ---------------------------------------------------------------------------------- tables
// tab8[aa] = { packed x^ (4 int8), |x1|^2 as float bits, L1(x^), 0 }; scale[0] = s, scale[1] = s^2/2
// tabR[aa] = { packed x^ of columns 0..3, packed x^ of columns 4..7 (own scale), |x|^2 over all 8
// columns as float bits, L1(x^ 0..3) | L1(x^ 4..7) << 16 } for the survivor refinement;
// scale[2] = s of columns 4..7, scale[3] = 1 when that table is usable
// tabW[aa] = { packed x^ of columns 0..3, of columns 4..7 (ONE scale over all 8 columns), |x|^2 over
// all 8 as float bits, L1(x^) over all 8 } for the wide rows; scale[4] = that s, scale[5] = s^2/2
__global__ void hs_jtables8_kernel(const double* __restrict__ coords, int alphabet,
                                   uint4* __restrict__ tab8, float* __restrict__ scale,
                                   uint32_t* __restrict__ unsafe, uint4* __restrict__ tabR,
                                   uint4* __restrict__ tabW) {
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
    // the constant factors of the gamma slots (last 16 bytes of a member's row) as data, at byte 768
    // of the table block: hs_join8x_kernel's lanes of the fourth quarter LOAD them in place of a record
    tab8[HS_J8_CONST_AT] = make_uint4(0x7f7f0000u, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x017f7f7fu);
  }
  // the other four columns, for hs_refine8_kernel
  __syncthreads();
  double m2 = 0.0;
  for (int j = QD; j < 8; ++j) m2 = fmax(m2, aa < alphabet ? fabs(coords[aa * 8 + j]) : 0.0);
  smax[aa] = m2;
  __syncthreads();
  double mm2 = 0.0;
  for (int i = 0; i < 32; ++i) mm2 = fmax(mm2, smax[i]);
  const bool ok2 = mm2 < 1e6;          // NaN or huge: no refinement
  const double mm2_raw = mm2;
  if (!(mm2 > 0.0) || !ok2) mm2 = 127.0;  // all-zero columns: any scale does
  const float s2f = (float)(127.0 / mm2);
  const double s2 = (double)s2f;
  uint32_t pack2 = 0;
  int l12 = 0;
  double n8 = n;
  for (int j = QD; j < 8; ++j) {
    const double v = aa < alphabet ? coords[aa * 8 + j] : 0.0;
    int q = (int)rint(s2 * v);
    q = max(-127, min(127, q));
    pack2 |= ((uint32_t)(q & 0xff)) << (8 * (j - QD));
    l12 += abs(q);
    n8 += v * v;
  }
  tabR[aa] = make_uint4(pack, pack2, __float_as_uint((float)n8), (uint32_t)l1 | ((uint32_t)l12 << 16));
  if (aa == 0) {
    scale[2] = s2f;
    scale[3] = ok2 ? 1.0f : 0.0f;
  }
  // all 8 columns on one scale (an unusable table was flagged above: mm covers columns 0..3, mm2
  // columns 4..7)
  const double mw = ok2 ? fmax(mm, mm2_raw) : mm;
  const double sw = 127.0 / mw;
  uint32_t pw[2] = {0u, 0u};
  int l1w = 0;
  for (int j = 0; j < 8; ++j) {
    const double v = aa < alphabet ? coords[aa * 8 + j] : 0.0;
    int q = (int)rint(sw * v);
    q = max(-127, min(127, q));
    pw[j >> 2] |= ((uint32_t)(q & 0xff)) << (8 * (j & 3));
    l1w += abs(q);
  }
  tabW[aa] = make_uint4(pw[0], pw[1], __float_as_uint((float)n8), (uint32_t)l1w);
  if (aa == 0) {
    scale[4] = (float)sw;
    scale[5] = (float)(0.5 * sw * sw);
    scale[6] = ok2 ? 1.0f : 0.0f;
  }
}

// ---------------------------------------------------------------------------------- query prep
// c8[q] (ROW = 32 KS bytes): byte 4p + j = c^ of coordinate j (< 4) of position p (< k), zeros up to
// byte ROW - 29; the last 28 bytes: (-127 x11, -1), 0, 0, then the 14 digits of -gamma.
template <bool WIDE>
__global__ __launch_bounds__(256) void hs_qprep8_kernel(const double* __restrict__ centers, uint32_t nq,
                                                        int k, double r2, const float* __restrict__ scale,
                                                        int8_t* __restrict__ c8,
                                                        uint32_t* __restrict__ unsafe,
                                                        int8_t* __restrict__ c8b) {
  const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;
  const int lane = lane_id();
  const int ROW = 32 * ks_of(k, WIDE), TAIL = ROW - 28;
  // WIDE: one row, all 8 columns on one scale, byte 8 pos + j
  const double sA = (double)scale[WIDE ? 4 : 0], sB = (double)scale[WIDE ? 4 : 2];
  const double* c = centers + (uint64_t)q * 8 * k;
  int8_t* outA = c8 + (uint64_t)q * ROW;
  int8_t* outB = c8b ? c8b + (uint64_t)q * ROW : nullptr;
  // one pass over the 8k doubles of the row (lane i, i + 64, ...: coalesced): coordinate j < 4 of
  // position p goes to byte 4p + j of the first row, coordinate j >= 4 to byte 4p + j - 4 of the
  // second; sums per half.  Positions k .. TAIL/4 - 1 are zero bytes.
  double ncA = 0.0, penA = 0.0, ncB = 0.0, penB = 0.0;
  int l1A = 0, l1B = 0;
  bool badA = false, badB = !(scale[3] > 0.f);
  for (int i = lane; i < (WIDE ? TAIL : 2 * TAIL); i += 64) {
    const int pos = i >> 3, j = i & 7;
    const bool second = !WIDE && j >= QD;
    int qv = 0;
    if (pos < k) {
      const double v = c[i];
      const double sv = (second ? sB : sA) * v;
      const bool bad = !(fabs(sv) < 1.0e6);
      // saturate; a coordinate outside +-127 has a quantisation error n_i > 1/2, which costs at
      // most (127 + 1/2)(|n_i| - 1/2) more in the bound (|x^_i| <= 127, |e_i| <= 1/2)
      qv = (int)fmax(-127.0, fmin(127.0, rint(sv)));
      const double pen = 127.5 * fmax(0.0, fabs(sv - (double)qv) - 0.5);
      if (second) {
        ncB += v * v;
        penB += pen;
        l1B += abs(qv);
        badB = badB || bad;
      } else {
        ncA += v * v;
        penA += pen;
        l1A += abs(qv);
        badA = badA || bad;
      }
    }
    if (WIDE) outA[i] = (int8_t)qv;
    else if (!second) outA[4 * pos + j] = (int8_t)qv;
    else if (outB) outB[4 * pos + j - QD] = (int8_t)qv;
  }
  for (int off = 32; off; off >>= 1) {
    ncA += __shfl_xor(ncA, off);
    penA += __shfl_xor(penA, off);
    l1A += __shfl_xor(l1A, off);
    ncB += __shfl_xor(ncB, off);
    penB += __shfl_xor(penB, off);
    l1B += __shfl_xor(l1B, off);
  }
  // gamma = floor(s^2 (nc - R^2)/2 - L1/2 - saturation penalty - 2)
  const double g = floor(0.5 * sA * sA * (ncA - r2) - 0.5 * (double)l1A - penA - 2.0);
  // a query far outside the table's range would make the filter uselessly permissive
  bool bad = badA || !(fabs(g) < 1.0e9) || !(penA < 30000.0);
  int d[DIG + 1];
  const int v = bad ? 0 : -(int)g;
  // -gamma too LARGE for the digits would have to be clamped in the non-permissive direction
  bool too_high = false;
  digits127<DIG>(v, d, &too_high);
  bad = bad || too_high;
  if (__ballot(bad) && lane == 0) atomicOr(unsafe, 1u);
  if (lane < 28) {
    int8_t b;
    if (lane < RDIG) b = (int8_t)-127;
    else if (lane == RDIG) b = (int8_t)-1;
    else if (lane < 14) b = 0;
    else b = (int8_t)d[lane - 14];
    outA[TAIL + lane] = b;
  }
  // Tail of the second row, for hs_refine8_kernel: byte ROW - 24 = the double |c|^2 - R^2 over all 8
  // columns, bytes ROW - 16 / ROW - 12 = the floats L1(c^)/2 + saturation penalty + 2 of columns
  // 0..3 / 4..7 (+inf: this query cannot be refined)
  if (outB) {
    const bool any_bad = __ballot(bad || badB) != 0;
    if (lane == 0) {
      *reinterpret_cast<double*>(outB + ROW - 24) = (ncA + ncB) - r2;
      const float inf = __builtin_inff();
      // rounded up (float): the bound may only get more permissive
      *reinterpret_cast<float*>(outB + ROW - 16) = any_bad ? inf : __double2float_ru(0.5 * (double)l1A + penA + 2.0);
      *reinterpret_cast<float*>(outB + ROW - 12) = any_bad ? inf : __double2float_ru(0.5 * (double)l1B + penB + 2.0);
    }
    if (lane == 1) {
      *reinterpret_cast<uint32_t*>(outB + ROW - 28) = 0u;
      *reinterpret_cast<uint64_t*>(outB + ROW - 8) = 0ull;
    }
  }
}

// The same two rows for a query that IS a k-mer of the coordinate table (self-join), from its codes:
// x^ = the table's quantised rows (what the members carry), no saturation, norms from the table's
// doubles.  One thread per query.
template <bool WIDE>
__global__ __launch_bounds__(256) void hs_qprep8_codes_kernel(const uint8_t* __restrict__ qcodes, uint32_t nq,
                                                              int k, double r2,
                                                              const double* __restrict__ coords,
                                                              const uint4* __restrict__ tab8,
                                                              const uint4* __restrict__ tabR,
                                                              const uint4* __restrict__ tabW,
                                                              const float* __restrict__ scale,
                                                              int8_t* __restrict__ c8, int8_t* __restrict__ c8b) {
  __shared__ uint32_t sA[32], sB[32], sL1[32];
  __shared__ double sNA[32], sNB[32];
  if (threadIdx.x < 32) {
    const int aa = threadIdx.x;
    // WIDE: both halves of a position's 8 bytes go to the one row; the L1 over all 8 counts as "A"
    sA[aa] = WIDE ? tabW[aa].x : tab8[aa].x;
    sB[aa] = WIDE ? tabW[aa].y : tabR[aa].y;
    sL1[aa] = WIDE ? tabW[aa].w : tabR[aa].w;  // L1(x^ 0..3) | L1(x^ 4..7) << 16
    double na = 0.0, nb = 0.0;
    for (int j = 0; j < 8; ++j) {
      const double v = coords[aa * 8 + j];
      if (j < QD) na += v * v; else nb += v * v;
    }
    sNA[aa] = na;
    sNB[aa] = nb;
  }
  __syncthreads();
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const int ROW = 32 * ks_of(k, WIDE), TAIL = ROW - 28;
  const uint8_t* code = qcodes + (uint64_t)q * k;
  uint32_t* outA = reinterpret_cast<uint32_t*>(c8 + (uint64_t)q * ROW);
  uint32_t* outB = (c8b && !WIDE) ? reinterpret_cast<uint32_t*>(c8b + (uint64_t)q * ROW) : nullptr;
  double ncA = 0.0, ncB = 0.0;
  uint32_t l1A = 0, l1B = 0;
  if constexpr (WIDE) {
    for (int p = 0; p < TAIL / 8; ++p) {
      uint32_t a = 0u, b = 0u;
      if (p < k) {
        const uint32_t c = code[p] & 31u;
        a = sA[c];
        b = sB[c];
        ncA += sNA[c] + sNB[c];
        l1A += sL1[c];
      }
      outA[2 * p] = a;
      outA[2 * p + 1] = b;
    }
    outA[TAIL / 4 - 1] = 0u;  // bytes 160..163
  } else
  for (int p = 0; p < TAIL / 4; ++p) {
    uint32_t a = 0u, b = 0u;
    if (p < k) {
      const uint32_t c = code[p] & 31u;
      a = sA[c];
      b = sB[c];
      ncA += sNA[c];
      ncB += sNB[c];
      l1A += sL1[c] & 0xffffu;
      l1B += sL1[c] >> 16;
    }
    outA[p] = a;
    if (outB) outB[p] = b;
  }
  const double sAs = (double)scale[WIDE ? 4 : 0];
  // gamma as in hs_qprep8_kernel (no saturation penalty: table values quantise inside +-127)
  const double g = floor(0.5 * sAs * sAs * (ncA - r2) - 0.5 * (double)l1A - 2.0);
  int d[DIG + 1];
  bool too_high = false;
  digits127<DIG>(-(int)g, d, &too_high);  // a k-mer of the table never overflows (hs_qprep8_kernel's
                                          // range checks are for arbitrary points); clamped if it did
  int8_t* tail = c8 + (uint64_t)q * ROW + TAIL;
  for (int i = 0; i < 28; ++i) tail[i] = i < RDIG ? (int8_t)-127 : i == RDIG ? (int8_t)-1 : i < 14 ? (int8_t)0 : (int8_t)d[i - 14];
  if (outB) {
    int8_t* tb = c8b + (uint64_t)q * ROW;
    *reinterpret_cast<uint32_t*>(tb + ROW - 28) = 0u;
    *reinterpret_cast<double*>(tb + ROW - 24) = (ncA + ncB) - r2;
    const bool okB = scale[3] > 0.f && !too_high;
    const float inf = __builtin_inff();
    *reinterpret_cast<float*>(tb + ROW - 16) = okB ? __double2float_ru(0.5 * (double)l1A + 2.0) : inf;
    *reinterpret_cast<float*>(tb + ROW - 12) = okB ? __double2float_ru(0.5 * (double)l1B + 2.0) : inf;
    *reinterpret_cast<uint64_t*>(tb + ROW - 8) = 0ull;
  }
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

// A k-mer's residues as ONE little-endian bit stream, 5 bits per position (position p at bit 5p):
// the packed form keeps 25 residues per 16-byte word (3 pad bits), so the second word is stitched
// on at bit 125.  v[0..7]; positions past the k-mer read as residue 0.
struct Stream256 {
  uint32_t v[8];
};
template <int PW>
__device__ __forceinline__ Stream256 stitch(const uint4 w0, const uint4 w1) {
  Stream256 s;
  s.v[0] = w0.x;
  s.v[1] = w0.y;
  s.v[2] = w0.z;
  if constexpr (PW == 1) {
    s.v[3] = w0.w;
    s.v[4] = s.v[5] = s.v[6] = s.v[7] = 0u;
  } else {
    s.v[3] = (w0.w & 0x1fffffffu) | (w1.x << 29);
    s.v[4] = __funnelshift_r(w1.x, w1.y, 3);
    s.v[5] = __funnelshift_r(w1.y, w1.z, 3);
    s.v[6] = __funnelshift_r(w1.z, w1.w, 3);
    s.v[7] = w1.w >> 3;
  }
  return s;
}
// residue at compile-time bit BIT of a stream (after an optional uniform shift of the whole stream)
template <int BIT>
__device__ __forceinline__ uint32_t stream_at(const Stream256& s) {
  constexpr int wi = BIT >> 5, sh = BIT & 31;
  if constexpr (wi > 7) return 0u;
  else if constexpr (sh > 27 && wi < 7) return __funnelshift_r(s.v[wi], s.v[wi + 1], sh) & 31u;
  else return (s.v[wi] >> sh) & 31u;
}

__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) {
  return ((uint32_t)a & 0xffu) | (((uint32_t)b & 0xffu) << 8) | (((uint32_t)c & 0xffu) << 16) |
         (((uint32_t)d & 0xffu) << 24);
}

// A operands (KS k-steps) of one 32-member row tile for lane (r, h): k-step s < KS - 1 carries
// positions 8s + 4h + {0,1,2,3} (positions >= k meet zero query bytes, so their rows need no
// masking); the last k-step = the entry's prebuilt record (h = 0) or the constant factors of the
// gamma slots (h = 1).
template <int KS, int PW>
__device__ __forceinline__ void build_afrags8(const uint4 pk, const uint4 pk1, const uint4 rec, int h,
                                              const uint32_t* sTab8, intx4 (&A)[KS]) {
  if constexpr (PW == 1) {
    // lanes of the upper half take positions 4..7, 12..15, ...: shift the word down by 20 bits
    const uint32_t sh = 20u * (uint32_t)h;
    const uint32_t x = __funnelshift_r(pk.x, pk.y, sh), y = __funnelshift_r(pk.y, pk.z, sh),
                   z = __funnelshift_r(pk.z, pk.w, sh), w = pk.w >> sh;
#define HS_A8(S, M) A[S][M] = (int)sTab8[residue_at<40 * S + 5 * M>(x, y, z, w)];
    HS_A8(0, 0) HS_A8(0, 1) HS_A8(0, 2) HS_A8(0, 3)
    HS_A8(1, 0) HS_A8(1, 1) HS_A8(1, 2) HS_A8(1, 3)
    HS_A8(2, 0) HS_A8(2, 1) HS_A8(2, 2) HS_A8(2, 3)
#undef HS_A8
  } else {
    Stream256 st = stitch<2>(pk, pk1);
    const uint32_t sh = 20u * (uint32_t)h;
#pragma unroll
    for (int i = 0; i < 7; ++i) st.v[i] = __funnelshift_r(st.v[i], st.v[i + 1], sh);
    st.v[7] >>= sh;
#define HS_A8(S, M) A[S][M] = (int)sTab8[stream_at<40 * (S) + 5 * (M)>(st)];
#define HS_A8S(S) HS_A8(S, 0) HS_A8(S, 1) HS_A8(S, 2) HS_A8(S, 3)
    HS_A8S(0) HS_A8S(1) HS_A8S(2) HS_A8S(3) HS_A8S(4)
    if constexpr (KS == 8) { HS_A8S(5) HS_A8S(6) }
#undef HS_A8S
#undef HS_A8
  }
  constexpr uint32_t C127 = 0x7f7f7f7fu;
  A[KS - 1][0] = h ? 0x7f7f0000 : (int)rec.x;          // bytes ROW-16, ROW-15 unused; ROW-14.. = 127
  A[KS - 1][1] = h ? (int)C127 : (int)rec.y;
  A[KS - 1][2] = h ? (int)C127 : (int)rec.z;
  A[KS - 1][3] = h ? 0x017f7f7f : (int)rec.w;          // last byte = 1 (remainder slot of -gamma)
}

// Wide rows (6 or 8 k-steps, all 8 coordinates): k-step s < KS - 1 carries positions 4 s + 2 h + {0, 1},
// 8 bytes each, by one 8-byte lookup per position (positions past the k-mer meet zero query bytes);
// the last k-step as above.
template <int KS>
__device__ __forceinline__ void build_afrags8_wide(const uint4 pk, const uint4 rec, int h, const uint2* sTabW,
                                                   intx4 (&A)[KS]) {
  const uint32_t sh = 10u * (uint32_t)h;  // the upper half's positions are two further on
  const uint32_t x = __funnelshift_r(pk.x, pk.y, sh), y = __funnelshift_r(pk.y, pk.z, sh),
                 z = __funnelshift_r(pk.z, pk.w, sh), w = pk.w >> sh;
#define HS_AW(S)                                                          \
  {                                                                       \
    const uint2 p0_ = sTabW[residue_at<20 * (S)>(x, y, z, w)];            \
    const uint2 p1_ = sTabW[residue_at<(20 * (S) + 5 < 124 ? 20 * (S) + 5 : 123)>(x, y, z, w)]; \
    A[S] = intx4{(int)p0_.x, (int)p0_.y, (int)p1_.x, (int)p1_.y};         \
  }
  HS_AW(0) HS_AW(1) HS_AW(2) HS_AW(3) HS_AW(4)
  if constexpr (KS == 8) { HS_AW(5) HS_AW(6) }
#undef HS_AW
  constexpr uint32_t C127 = 0x7f7f7f7fu;
  A[KS - 1][0] = h ? 0x7f7f0000 : (int)rec.x;
  A[KS - 1][1] = h ? (int)C127 : (int)rec.y;
  A[KS - 1][2] = h ? (int)C127 : (int)rec.z;
  A[KS - 1][3] = h ? 0x017f7f7f : (int)rec.w;
}

// index build: bucket-ordered packed copy of one table + the 16-byte A-row tail of every entry
__global__ __launch_bounds__(256) void hs_gather_rec8_kernel(const uint4* __restrict__ packed_all,
                                                             const uint32_t* __restrict__ ids,
                                                             uint32_t n, int k, int PW, int wide,
                                                             const uint4* __restrict__ tab8,
                                                             const uint4* __restrict__ tabW,
                                                             const float* __restrict__ scale,
                                                             uint4* __restrict__ out_packed,
                                                             uint4* __restrict__ out_rec,
                                                             uint32_t* __restrict__ out_rho) {
  // rows {x^, |x|^2 (float bits), L1(x^)}: of the 4 filter columns, or (wide) of all 8
  __shared__ uint4 sTab[32];
  if (threadIdx.x < 32)
    sTab[threadIdx.x] = wide ? make_uint4(0u, tabW[threadIdx.x].z, tabW[threadIdx.x].w, 0u) : tab8[threadIdx.x];
  __syncthreads();
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  // the position whose x^ opens the record (if the k-mer has it; never with wide rows)
  const int plast = wide ? 1 << 20 : 8 * ks_of(k) - 8;
  double nx = 0.0;
  int l1 = 0;
  uint32_t xlast = 0, rlast = 32u;  // (32: the k-mer has no residue at that position)
  for (int wd = 0; wd < PW; ++wd) {
    const uint4 pk = packed_all[(uint64_t)ids[t] * PW + wd];
    if (out_packed) out_packed[(uint64_t)t * PW + wd] = pk;  // (null: records only)
    const uint32_t w[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
    for (int r = 0; r < 25; ++r) {
      const int bit = 5 * r, wi = bit >> 5, sh = bit & 31;
      uint32_t c = w[wi] >> sh;
      if (sh > 27) c |= w[wi + 1] << (32 - sh);
      const uint4 row = sTab[c & 31u];
      const int p = 25 * wd + r;
      if (p < k) {
        nx += (double)__uint_as_float(row.y);
        l1 += (int)row.z;
        if (p == plast) {
          xlast = row.x;
          rlast = c & 31u;
        }
      }
    }
  }
  // rho = floor(s^2 |x1|^2 / 2 - L1(x^)/2 - dims/4 - 2); the 2 absorbs the fp32 roundings of
  // scale[1] and of the table's squared norms
  const double rho = floor((double)scale[wide ? 5 : 1] * nx - 0.5 * (double)l1 -
                           0.25 * (double)((wide ? 8 : QD) * k) - 2.0);
  int d[RDIG + 1];
  digits127<RDIG>((int)rho, d, nullptr);  // too large -> clamped down: more permissive, never less
  out_rec[t] = make_uint4(xlast, pack4(d[0], d[1], d[2], d[3]), pack4(d[4], d[5], d[6], d[7]),
                          pack4(d[8], d[9], d[10], d[11]));
  // The same record in FOUR bytes, for hs_join8r_kernel (which is bound by the bytes it reads per member):
  // the digits are a function of q = rho div 127 alone -- digit j = clamp(|q| - 127 j, 0, 127), negated for
  // q < 0 -- and come from a table there; x^ of the record's position from the residue.  Bits 0..10: |q|,
  // 11: q < 0, 12..18: rho mod 127, 20..25: that residue (32: none), 26..31: zero -- the residue is read as the
  // TEN-bit field at bit 20 the other rows of the operand build read there (join8r_lookup).
  if (out_rho) {
    int q = (int)rho / 127, rem = (int)rho - q * 127;  // as digits127
    if (rem < 0) {
      rem += 127;
      q -= 1;
    }
    if (q > 127 * RDIG) {
      q = 127 * RDIG;
      rem = 126;
    }
    if (q < -127 * RDIG) {
      q = -127 * RDIG;
      rem = 0;
    }
    out_rho[t] = (uint32_t)abs(q) | (q < 0 ? 0x800u : 0u) | ((uint32_t)rem << 12) | (rlast << 20);
  }
}

#ifdef HS_JOIN_TIMING
// s_memtime ordered after the value `dep` exists (a scalar), so an interval really ends when the
// work producing dep has completed
__device__ __forceinline__ uint64_t memtime_after(uint32_t dep) {
  uint64_t t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(dep) : "memory");
  return t;
}
#define HS_TD(i, dep) { const uint64_t now_ = memtime_after((uint32_t)(dep)); tacc[i] += now_ - tlast; tlast = now_; }
#define HS_T(i) HS_TD(i, 0)
__device__ unsigned long long g_join8_timing[8];
#else
#define HS_T(i)
#define HS_TD(i, dep)
#endif

// --------------------------------------------------------------------------------------- join
// Query rows in TILE-FRAGMENT order: the 32-query tile starting at segment-order row p0 (a multiple
// of 32 inside its segment), nr = rows of the tile (32, less at the segment's ragged end), keeps
// 16-byte piece g = 2 s + h of its row j at uint4 index p0 * 8 + g * nr + j: the B operand of
// k-step s is then ONE fully coalesced 16-byte-per-lane load, straight into registers.
__global__ __launch_bounds__(256) void hs_gather_c8t_kernel(const int8_t* __restrict__ c8,
                                                            const uint32_t* __restrict__ sorted_ql,
                                                            const uint32_t* __restrict__ seg_qoff,
                                                            const uint32_t* __restrict__ seg_of,
                                                            uint32_t nql, int L, int pieces,
                                                            uint4* __restrict__ out) {
  // EIGHT lanes per probe, lane g of them moving the pieces g, g + 8, ...: a load instruction of the wave then
  // reads 8 whole rows (8 cache lines) instead of one piece of 64 different rows (64 lines) -- these kernels
  // are bound by the lines their loads and stores touch, not by the bytes
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t p = t >> 3, g0 = t & 7u;
  if (p >= nql) return;
  const uint32_t lo = seg_of[p];  // the segment of sorted position p (written when the probes were grouped)
  const uint32_t qoff = seg_qoff[lo], nQ = seg_qoff[lo + 1] - qoff;
  const uint32_t local = p - qoff, tile = local >> 5, j = local & 31u;
  const uint32_t nr = min(32u, nQ - (tile << 5));
  const uint32_t q = sorted_ql[p] / (uint32_t)L;
  const uint4* src = reinterpret_cast<const uint4*>(c8) + (uint64_t)q * pieces;
  uint4* dst = out + (uint64_t)(qoff + (tile << 5)) * pieces + j;
  for (uint32_t g = g0; g < (uint32_t)pieces; g += 8u) dst[g * nr] = src[g];
}

// No LDS staging and no workgroup barrier: a work item is 128 bucket members x <= 2048 probing
// queries and belongs to ONE wave, which holds the members' A operands in registers and streams the
// B tiles from L2 straight into registers (double-buffered, one tile ahead, across item
// boundaries).  Waves take their first item by position and every further one from a global
// counter (two items ahead, so the descriptor and the packed members are there in time): a stall
// -- survivor bookkeeping, a late load -- costs that wave only, and no wave idles at the end.
template <int KS>
__device__ __forceinline__ void load_btile(intx4 (&B)[KS], const uint4* __restrict__ c8t, uint32_t row0,
                                           uint32_t nr, int lane) {
  const uint4* base = c8t + (uint64_t)row0 * (2 * KS);
  if (nr == 32u) {  // full tile: piece (s, h) of row r sits at 64 s + lane
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const uint4 v = base[64 * s + lane];
      B[s] = intx4{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
    }
  } else {  // ragged last tile of a segment: nr rows, piece g of row j at g * nr + j
    const uint32_t r = (uint32_t)lane & 31u, h = (uint32_t)lane >> 5;
    const uint32_t j = min(r, nr - 1u) + h * nr;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const uint4 v = base[(uint32_t)(2 * s) * nr + j];
      B[s] = intx4{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
    }
  }
}

// Survivors of one accumulator group (GT row tiles of 32 members, first one T0) against the 32
// queries of the tile at segment-relative offset qc: rows (i & 3) + 8 (i >> 2) + 4 h of tile t with
// acc >= 0.  Rare path; slots of the survivor list are reserved 64 at a time per wave.
template <int GT>
__device__ __forceinline__ void emit_survivors(const intx16 (&acc)[GT], int T0, uint32_t qc,
                                               uint32_t qoff, uint32_t q_end, uint32_t wbase,
                                               uint32_t M, uint32_t mstart, int lane,
                                               uint32_t& res_base, uint32_t& res_used,
                                               uint32_t* __restrict__ prov_count, uint32_t prov_cap,
                                               uint2* __restrict__ prov) {
  const int r = lane & 31, h = lane >> 5;
  const bool col_ok = qc + (uint32_t)r < q_end;
  const uint32_t ql = HS_PROV_INDIRECT | (qoff + qc + (uint32_t)r);
#pragma unroll
  for (int t = 0; t < GT; ++t) {
    uint32_t any = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 16; ++i) any &= (uint32_t)acc[t][i];
    if (!__ballot((int)any >= 0)) continue;  // no survivor in this row tile
    uint32_t neg = 0;  // bit i = sign of acc[t][i]: shift the sign bits in, last element first
#pragma unroll
    for (int i = 15; i >= 0; --i) neg = __builtin_amdgcn_alignbit(neg, (uint32_t)acc[t][i], 31);
    uint32_t mask = col_ok ? (~neg & 0xffffu) : 0u;
    while (__ballot(mask != 0)) {
      uint32_t idx = 0;
      bool pass = false;
      if (mask) {
        const int i = __ffs((int)mask) - 1;
        mask &= mask - 1;
        idx = wbase + (uint32_t)((T0 + t) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h);
        pass = idx < M;
      }
      const unsigned long long m = __ballot(pass);
      if (m) {
        const uint32_t cnt = (uint32_t)__popcll(m);
        if (res_used + cnt > JRES) {
          close_reservation(prov, res_base, res_used, prov_cap, lane);
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

// sign bit of the result = AND of the sign bits of the accumulators of a GT-tile group
template <int GT>
__device__ __forceinline__ uint32_t and_tree(const intx16 (&acc)[GT]) {
  uint32_t a = 0xffffffffu;
#pragma unroll
  for (int t = 0; t < GT; ++t)
#pragma unroll
    for (int i = 0; i < 16; i += 2) a = a & (uint32_t)acc[t][i] & (uint32_t)acc[t][i + 1];
  return a;
}

// wave-uniform copy of a descriptor word (keeps it, and everything derived from it, in SGPRs)
__device__ __forceinline__ uint4 uniform4(const uint4 v) {
  return make_uint4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y),
                    __builtin_amdgcn_readfirstlane(v.z), __builtin_amdgcn_readfirstlane(v.w));
}

// The wave's JT row tiles form two accumulator groups X (the first JT/2 tiles) and Y (the rest).  Per
// query tile: 8 MFMAs into X while the sign test of Y (previous query tile) issues in their gaps,
// then 8 MFMAs into Y beside the sign test of X -- the vector instructions of the epilogue never
// stand between two MFMAs of the same wave.
template <int JT, int KS, bool WIDE = false>
__global__ __launch_bounds__(256, 2) void hs_join8w_kernel(
    const uint4* __restrict__ desc, uint32_t n_items, const uint4* __restrict__ packed_base,
    const uint4* __restrict__ rec_base, const uint4* __restrict__ c8t,
    const uint4* __restrict__ tab8, uint32_t* __restrict__ prov_count, uint32_t prov_cap,
    uint2* __restrict__ prov, uint32_t* __restrict__ item_counter, uint32_t G,
    const uint32_t* __restrict__ n_items_dev) {
  // n_items = the capacity of desc; the real count may only be known on the device (no host round
  // trip between cutting the items and joining them)
  if (n_items_dev) n_items = min(n_items, __builtin_amdgcn_readfirstlane(*n_items_dev));
  static_assert(JT == 4 || JT == 2, "two accumulator groups of JT / 2 row tiles");
  static_assert(KS == 4 || KS == 6 || KS == 8, "k-steps of a row");
  static_assert(!WIDE || KS == 6 || KS == 8, "wide rows have 6 or 8 k-steps");
  constexpr int GT = JT / 2;
  constexpr int PW = (KS == 4 || WIDE) ? 1 : 2;  // packed words per member
  __shared__ uint32_t sTab8[32];
  __shared__ uint2 sTabW[WIDE ? 32 : 1];
  const int tid = threadIdx.x, lane = tid & 63;
  // wave-uniform by construction: say so, or every per-item quantity derived from it (descriptor
  // addresses, loop bounds) is computed per lane and the descriptor loads become vector loads
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
#ifdef HS_JOIN_TIMING
  uint64_t tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
  const uint64_t tstart = tlast;
#endif
  if (tid < 32) {
    sTab8[tid] = tab8[tid].x;
    if constexpr (WIDE) sTabW[tid] = make_uint2(tab8[tid].x, tab8[tid].y);  // (the caller passes tabW)
  }
  __syncthreads();  // the only one: the table is read-only from here on
  // Items come in chunks of G consecutive ones (same-address atomics are slow: one per chunk): the
  // first chunk by position, every further one from the counter, requested a whole chunk ahead.
  // Descriptors are fetched TWO items ahead (scalar loads), so the next item's is in registers
  // when its member loads are issued.
  const uint32_t first_dynamic = gridDim.x * 4u * G;
  uint32_t item = (blockIdx.x * 4u + (uint32_t)wave) * G;
  if (item >= n_items) return;
  uint32_t res_base = 0, res_used = JRES;
  uint32_t next_chunk_v = 0;
  if (lane == 0) next_chunk_v = atomicAdd(item_counter, G);
  // the item two ahead of the current one, and the end of the chunk it belongs to
  uint32_t pf_item = item, pf_chunk_end = item + G;
#define HS_ADVANCE_PF()                                                                  \
  {                                                                                      \
    ++pf_item;                                                                           \
    if (pf_item == pf_chunk_end) {                                                       \
      pf_item = first_dynamic + __builtin_amdgcn_readfirstlane(next_chunk_v);            \
      pf_chunk_end = pf_item + G;                                                        \
      if (lane == 0 && pf_item < n_items) next_chunk_v = atomicAdd(item_counter, G);     \
    }                                                                                    \
  }
  uint4 d0 = uniform4(desc[2 * (uint64_t)item]), d1 = uniform4(desc[2 * (uint64_t)item + 1]);
  HS_ADVANCE_PF()
  uint32_t next_item = pf_item;
  uint4 nd0 = d0, nd1 = d1;
  if (next_item < n_items) {
    nd0 = uniform4(desc[2 * (uint64_t)next_item]);
    nd1 = uniform4(desc[2 * (uint64_t)next_item + 1]);
  }
  uint4 mk[JT];                  // lanes of the lower half: packed member, upper half: its record
  uint4 mk1[PW == 2 ? JT : 1];   // second packed word of the member (k > 25)
  constexpr int NB = 3;          // B tiles in flight per wave: the one in use + two prefetched
  constexpr uint32_t GQ = 32 * NB;  // queries per group of NB tiles
  intx4 Bq[NB][KS];
  // one 16-byte load per lane and row tile: the lower half-wave fetches the packed members, the
  // upper half their records; the halves trade them at build time (v_permlane32_swap)
#define HS_LOAD_MEMBERS(D0)                                                              \
  {                                                                                      \
    /* off_ = entry number of the segment's first member (record index; packed index / PW) */ \
    const int64_t off_ = (int64_t)(((uint64_t)(D0).y << 32) | (uint64_t)(D0).x);         \
    const uint32_t idx_ = (D0).w * (32 * JT) + r;                                        \
    _Pragma("unroll") for (int t = 0; t < JT; ++t) {                                     \
      const int64_t e_ = off_ + (int64_t)min(idx_ + 32 * t, (D0).z - 1);                 \
      mk[t] = h ? rec_base[e_] : packed_base[e_ * PW];                                   \
      if constexpr (PW == 2) mk1[t] = packed_base[e_ * PW + 1];                          \
    }                                                                                    \
  }
  // Waves start an item's query tiles at a wave-dependent pair and wrap around: resident waves
  // work on neighbouring items of the same (bucket, query group); walking the tiles in lock step
  // would make every tile a simultaneous first touch.
  const uint32_t skew = ((blockIdx.x * 4u + (uint32_t)wave) * 40503u) & 0xffffu;
#define HS_N_GROUPS(D1) (((D1).z - (D1).y + GQ - 1u) / GQ)
#define HS_FIRST_Q(D1) ((D1).y + GQ * ((skew * HS_N_GROUPS(D1)) >> 16))
  // Every B prefetch below is UNCONDITIONAL (a tile that does not exist is replaced by a harmless
  // valid one): the vector-memory counter is waited on by count, in issue order, so the number of
  // loads in flight at each wait must not depend on the path taken.
#define HS_LOAD_GROUP(ROW, Q0, QEND)                                                             \
  _Pragma("unroll") for (int u = 0; u < NB; ++u) {                                               \
    const uint32_t qu_ = (Q0) + 32u * u < (QEND) ? (Q0) + 32u * u : (Q0);                        \
    load_btile(Bq[u], c8t, (ROW) + qu_, min(32u, (QEND) - qu_), lane);                           \
  }
  HS_LOAD_MEMBERS(d0)
  {
    const uint32_t q0 = HS_FIRST_Q(d1);
    HS_LOAD_GROUP(d1.x, q0, d1.z)
  }
  while (true) {
    const uint32_t M = d0.z, mt = d0.w;
    const uint32_t qoff = d1.x, q_begin = d1.y, q_end = d1.z, mstart = d1.w;
    const uint32_t wbase = mt * (32 * JT);
    const bool has_next = next_item < n_items;
    HS_ADVANCE_PF()  // pf_item = the item after next
    HS_T(0)          // chunk bookkeeping
    uint4 nnd0 = nd0, nnd1 = nd1;
    if (pf_item < n_items) {
      nnd0 = uniform4(desc[2 * (uint64_t)pf_item]);
      nnd1 = uniform4(desc[2 * (uint64_t)pf_item + 1]);
    }
    intx4 A[JT][KS];
#pragma unroll
    for (int t = 0; t < JT; ++t) {
      uint4 pk, rk;  // after the swap: pk = the lower half's value, rk = the upper half's, in all lanes
#define HS_SWAP(C)                                                                        \
  {                                                                                       \
    const auto sw_ = __builtin_amdgcn_permlane32_swap(mk[t].C, mk[t].C, false, false);    \
    pk.C = sw_[0];                                                                        \
    rk.C = sw_[1];                                                                        \
  }
      HS_SWAP(x) HS_SWAP(y) HS_SWAP(z) HS_SWAP(w)
#undef HS_SWAP
      uint4 pk1 = pk;
      if constexpr (PW == 2) pk1 = mk1[t];
      if constexpr (WIDE) build_afrags8_wide<KS>(pk, rk, h, sTabW, A[t]);
      else build_afrags8<KS, PW>(pk, pk1, rk, h, sTab8, A[t]);
    }
    HS_TD(1, __builtin_amdgcn_readfirstlane(A[0][0][0] ^ A[JT - 1][3][3] ^ A[JT - 1][2][0] ^ A[1][1][1]))
    HS_LOAD_MEMBERS(nd0)
    HS_T(2)  // descriptor of the next item + issue of its member loads
    intx16 accX[GT], accY[GT];
#pragma unroll
    for (int t = 0; t < GT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) accY[t][i] = -1;  // "no survivor" for the first tile's Y test
    uint32_t prev_qc = q_begin;
    const uint32_t n_groups = HS_N_GROUPS(d1);
    uint32_t qc0 = HS_FIRST_Q(d1);
    // one group of NB query tiles (the later ones may not exist); the first group of an item is a
    // copy of its own, outside the loop, because the member loads just issued change the counts
    auto do_group = [&](uint32_t gi) {
      // the group after this one: of this item (wrapping around), or the next item's first
      uint32_t nrow = qoff, nq0 = qc0 + GQ, nqend = q_end;
      if (nq0 >= q_end) nq0 = q_begin;
      if (gi + 1 == n_groups) {
        nrow = nd1.x;
        nq0 = HS_FIRST_Q(nd1);
        nqend = nd1.z;
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const uint32_t qc = qc0 + 32u * (uint32_t)u;
        intx4 (&B)[KS] = Bq[u];
        if (u == 0 || qc < q_end) {
          // ---- phase 1: X <- A[0..GT) x B, beside the sign test of Y (previous query tile)
#pragma unroll
          for (int t = 0; t < GT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) accX[t][i] = 0;
          const uint32_t sY = and_tree<GT>(accY);
#pragma unroll
          for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < GT; ++t)
              accX[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[t][s], B[s], accX[t], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < KS * GT; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          }
          if (__ballot((int)sY >= 0))
            emit_survivors<GT>(accY, GT, prev_qc, qoff, q_end, wbase, M, mstart, lane, res_base,
                               res_used, prov_count, prov_cap, prov);
          // ---- phase 2: Y <- A[GT..JT) x B, beside the sign test of X
#pragma unroll
          for (int t = 0; t < GT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) accY[t][i] = 0;
#pragma unroll
          for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < GT; ++t)
              accY[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[GT + t][s], B[s], accY[t], 0, 0, 0);
          const uint32_t sX = and_tree<GT>(accX);
#pragma unroll
          for (int g = 0; g < KS * GT; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          }
          if (__ballot((int)sX >= 0))
            emit_survivors<GT>(accX, 0, qc, qoff, q_end, wbase, M, mstart, lane, res_base, res_used,
                               prov_count, prov_cap, prov);
          prev_qc = qc;
        }
        // the same-numbered tile of the next group into the registers just consumed
        const uint32_t nb = nq0 + 32u * (uint32_t)u < nqend ? nq0 + 32u * (uint32_t)u : nq0;
        load_btile(B, c8t, nrow + nb, min(32u, nqend - nb), lane);
      }
      qc0 = nq0;
    };
    do_group(0);
    HS_T(3)
    for (uint32_t gi = 1; gi < n_groups; ++gi) do_group(gi);
    HS_T(4)
    {  // the item's last Y group
      const uint32_t sY = and_tree<GT>(accY);
      if (__ballot((int)sY >= 0))
        emit_survivors<GT>(accY, GT, prev_qc, qoff, q_end, wbase, M, mstart, lane, res_base, res_used,
                           prov_count, prov_cap, prov);
    }
    HS_T(5)
    if (!has_next) break;
    item = next_item;
    next_item = pf_item;
    d0 = nd0;
    d1 = nd1;
    nd0 = nnd0;
    nd1 = nnd1;
  }
#undef HS_ADVANCE_PF
#undef HS_LOAD_GROUP
#undef HS_N_GROUPS
#undef HS_LOAD_MEMBERS
#undef HS_FIRST_Q
  close_reservation(prov, res_base, res_used, prov_cap, lane);
#ifdef HS_JOIN_TIMING
  HS_T(6)
  if (lane == 0) {
    for (int i = 0; i < 7; ++i) atomicAdd(&g_join8_timing[i], (unsigned long long)tacc[i]);
    atomicMax(&g_join8_timing[7], (unsigned long long)(tlast - tstart));
  }
#endif
}

// ------------------------------------------------------------------------ join, 16x16x64 shape
// The same work items, operands and filter value on v_mfma_i32_16x16x64_i8 (k <= 25).  Both shapes
// take the same cycles per operation; the chip, which lowers its clock under this load, holds a
// higher clock on the 16x16 shape: tools/ubench/mfma_shape.hip measures 2.7-3.1 POP/s for the
// 32x32x32 loop on random operands whether the pipe is 100 %, 70 % or 55 % busy (power, not issue
// slots, is the limit) and 3.2-3.6 POP/s for the 16x16x64 loop at 100 % and 70 %.
//
// Operand layout (checked by tools/ubench/mfma16_i8_layout.hip): lane (n = lane & 15, q = lane >> 4)
// holds row / column n, bytes 64 s + 16 q .. + 15 of k-step s (two k-steps of 64 bytes); result
// register i of that lane = row 4 q + i, column n.  A wave's 128 members are 8 row tiles of 16; a
// 32-query tile is two column tiles.  Per query tile: group X = row tiles 0..3 (16 MFMAs), group
// Y = row tiles 4..7, the sign test of one group in the gaps of the other group's MFMAs, as in
// hs_join8w_kernel.  In k-step 1 the lanes q = 0, 1 carry positions 16..23, q = 2 the member's
// record (position 24 + the rho digits), q = 3 the constant factors of the gamma slots.  The query
// rows are read from the same tile-fragment array (piece g of row j of a tile with nr rows at
// g nr + j): lane (n, q) takes piece 4 s + q of row 16 c + n for column tile c.
typedef int intx4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void load_btile_x(intx4 (&B)[2][2], const uint4* __restrict__ c8t, uint32_t row0,
                                             uint32_t nr, int lane) {
  // piece 4 s + q of row 16 c + n of a tile with nr rows sits at (4 s + q) nr + row.  One code path
  // for full and ragged tiles (rows past the end repeat the last one; masked later): a wave-uniform
  // base per k-step (scalar registers) plus two per-lane byte offsets -- the loads are of the
  // scalar-base + vector-offset form and their number never depends on the tile
  const char* t0 = reinterpret_cast<const char*>(c8t + (uint64_t)row0 * 8);
  const char* t1 = t0 + (uint64_t)nr * 64u;
  const uint32_t n = (uint32_t)lane & 15u, q = (uint32_t)lane >> 4;
  const uint32_t qn = q * nr;
  const uint32_t o0 = (qn + min(n, nr - 1u)) * 16u, o1 = (qn + min(16u + n, nr - 1u)) * 16u;
  const uint4 v00 = *reinterpret_cast<const uint4*>(t0 + o0), v01 = *reinterpret_cast<const uint4*>(t0 + o1);
  const uint4 v10 = *reinterpret_cast<const uint4*>(t1 + o0), v11 = *reinterpret_cast<const uint4*>(t1 + o1);
  B[0][0] = intx4{(int)v00.x, (int)v00.y, (int)v00.z, (int)v00.w};
  B[0][1] = intx4{(int)v01.x, (int)v01.y, (int)v01.z, (int)v01.w};
  B[1][0] = intx4{(int)v10.x, (int)v10.y, (int)v10.z, (int)v10.w};
  B[1][1] = intx4{(int)v11.x, (int)v11.y, (int)v11.z, (int)v11.w};
}

// sign bit of the result = AND of the sign bits of a group's 4 x NCOL accumulator tiles
template <int NCOL = 2>
__device__ __forceinline__ uint32_t and_tree_x(const intx4 (&acc)[4][2]) {
  uint32_t a = 0xffffffffu;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < NCOL; ++c)
      a = a & (uint32_t)acc[t][c][0] & (uint32_t)acc[t][c][1] & (uint32_t)acc[t][c][2] & (uint32_t)acc[t][c][3];
  return a;
}

// Survivors of one group (row tiles T0 .. T0 + 3) against the 16 NCOL queries at segment-relative offset qc
template <int NCOL = 2>
__device__ __forceinline__ void emit_survivors_x(const intx4 (&acc)[4][2], int T0, uint32_t qc, uint32_t qoff,
                                                 uint32_t q_end, uint32_t wbase, uint32_t M, uint32_t mstart,
                                                 int lane, uint32_t& res_base, uint32_t& res_used,
                                                 uint32_t* __restrict__ prov_count, uint32_t prov_cap,
                                                 uint2* __restrict__ prov) {
  const uint32_t n = (uint32_t)lane & 15u, q = (uint32_t)lane >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
      const uint32_t any = (uint32_t)acc[t][c][0] & (uint32_t)acc[t][c][1] & (uint32_t)acc[t][c][2] &
                           (uint32_t)acc[t][c][3];
      if (!__ballot((int)any >= 0)) continue;  // no survivor in this 16 x 16 tile
      uint32_t mask = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) mask |= (acc[t][c][i] >= 0 ? 1u : 0u) << i;
      const uint32_t col = qc + 16u * (uint32_t)c + n;
      if (!(col < q_end)) mask = 0u;
      const uint32_t ql = HS_PROV_INDIRECT | (qoff + col);
      while (__ballot(mask != 0)) {
        uint32_t idx = 0;
        bool pass = false;
        if (mask) {
          const int i = __ffs((int)mask) - 1;
          mask &= mask - 1;
          idx = wbase + (uint32_t)(16 * (T0 + t)) + 4u * q + (uint32_t)i;
          pass = idx < M;
        }
        const unsigned long long m = __ballot(pass);
        if (m) {
          const uint32_t cnt = (uint32_t)__popcll(m);
          if (res_used + cnt > JRES) {
            close_reservation(prov, res_base, res_used, prov_cap, lane);
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

__global__ __launch_bounds__(256, 2) void hs_join8x_kernel(
    const uint4* __restrict__ desc, uint32_t n_items, const uint4* __restrict__ packed_base,
    const uint4* __restrict__ rec_base, const uint4* __restrict__ c8t,
    const uint4* __restrict__ tab8, uint32_t* __restrict__ prov_count, uint32_t prov_cap,
    uint2* __restrict__ prov, uint32_t* __restrict__ item_counter, uint32_t G,
    const uint32_t* __restrict__ n_items_dev, uint32_t xcd_run) {
  if (n_items_dev) n_items = min(n_items, __builtin_amdgcn_readfirstlane(*n_items_dev));
  constexpr int RT = 8;  // row tiles of 16 members per wave
  // x^ of TWO consecutive residues per lookup: entry (r1 << 5 | r0) = {x^(r0), x^(r1)} -- 8 KB of LDS,
  // half the extractions and LDS reads of the operand build (4 ten-bit fields per row tile and lane)
  __shared__ uint2 sPair[1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4, up = lane >> 5;
  for (int e = tid; e < 1024; e += 256) sPair[e] = make_uint2(tab8[e & 31].x, tab8[e >> 5].x);
  __syncthreads();  // the only one: the table is read-only from here on
#ifdef HS_JOIN_TIMING
  uint64_t tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
  const uint64_t tstart = tlast;
#endif
  // Two ways of dealing the chunks (G consecutive items) to the waves.  xcd_run == 0: the first chunk by
  // wave number, the rest from ONE counter -- consecutive chunks go to whichever wave asks next, on any XCD.
  // xcd_run > 0: the chunk sequence is cut into runs of xcd_run chunks, run r belongs to XCD r % 8, and a wave
  // takes from its own XCD's counter (item_counter[8 + x]); consecutive items are the member tiles of ONE
  // segment and all of them stream that segment's query tiles, so a run's waves find them in their XCD's L2
  // instead of every XCD fetching every segment's tiles over the fabric.  A wave whose XCD has run dry
  // takes from the next XCD's counter (and stays there): the tail is dealt like the single counter's.
  // (xcd_run = a power of two.  The wave's XCD, the victim and the number of XCDs found dry are wave-uniform
  // and live in scalar registers: this kernel has no vector register to spare.)
  const uint32_t xcd_sh = 31u - (uint32_t)__builtin_clz(xcd_run | 1u), xcd_mask = (1u << xcd_sh) - 1u;
  uint32_t victim = __builtin_amdgcn_s_getreg(6164) & 7u;  // hwreg(HW_REG_XCC_ID, 0, 4)
  uint32_t tried = 0;
  const uint32_t first_dynamic = gridDim.x * 4u * G;
  // counter value v (already taken) -> the first item of the chunk it stands for; uniform
  auto chunk_item = [&](uint32_t v) -> uint32_t {
    if (!xcd_run) return first_dynamic + v;
    for (;;) {
      const uint32_t c = ((((v >> xcd_sh) << 3) + victim) << xcd_sh) + (v & xcd_mask);
      const uint64_t it = (uint64_t)c * G;
      if (it < (uint64_t)n_items) return (uint32_t)it;
      if (++tried == 8u) return 0xf0000000u;  // >= any n_items, and a few increments away from wrapping
      victim = (victim + 1u) & 7u;
      uint32_t j = 0;
      if (lane == 0) j = atomicAdd(item_counter + 8 + victim, 1u);
      v = __builtin_amdgcn_readfirstlane(j);
    }
  };
#define HS_TAKE_CHUNK() (xcd_run ? atomicAdd(item_counter + 8 + victim, 1u) : atomicAdd(item_counter, G))
  uint32_t item = (blockIdx.x * 4u + (uint32_t)wave) * G;
  if (xcd_run) {
    uint32_t first = 0;
    if (lane == 0) first = HS_TAKE_CHUNK();
    item = chunk_item(__builtin_amdgcn_readfirstlane(first));
  }
  if (item >= n_items) return;
  uint32_t res_base = 0, res_used = JRES;
  uint32_t next_chunk_v = 0;
  if (lane == 0) next_chunk_v = HS_TAKE_CHUNK();
  // A chunk ENDS at the last item there is: the wave leaves its loop at the first item number past the end,
  // and with XCD-local runs the chunk after the one that straddles the end of the list may well be a valid one
  // of another XCD's -- a wave that walked the straddling chunk's empty tail took that chunk (and the one
  // behind it, ahead of time) and left with both unprocessed whenever the tail was shorter than its look-ahead
  // of two items (G = 48 at 7 121 711 items: 96 items of 7 million never joined, a hit or two in a million lost).
  uint32_t pf_item = item, pf_chunk_end = min(item + G, n_items);
#define HS_ADVANCE_PF()                                                                  \
  {                                                                                      \
    ++pf_item;                                                                           \
    if (pf_item == pf_chunk_end) {                                                       \
      pf_item = chunk_item(__builtin_amdgcn_readfirstlane(next_chunk_v));                \
      pf_chunk_end = pf_item < n_items ? min(pf_item + G, n_items) : pf_item + G;        \
      if (lane == 0 && pf_item < n_items) next_chunk_v = HS_TAKE_CHUNK();                \
    }                                                                                    \
  }
  uint4 d0 = uniform4(desc[2 * (uint64_t)item]), d1 = uniform4(desc[2 * (uint64_t)item + 1]);
  HS_ADVANCE_PF()
  uint32_t next_item = pf_item;
  uint4 nd0 = d0, nd1 = d1;
  if (next_item < n_items) {
    nd0 = uniform4(desc[2 * (uint64_t)next_item]);
    nd1 = uniform4(desc[2 * (uint64_t)next_item + 1]);
  }
  // lanes 0..31: packed member 16 t + n; lanes 32..47: its record; lanes 48..63: the constant factors
  // of the gamma slots (one 16-byte block for everybody: their index is pinned to the segment's last
  // entry and their base moved back by as much, so the same clamped-index load serves all lanes)
  uint4 mk[RT];
  constexpr int NB = 3;          // B tiles in flight per wave: the one in use + two prefetched
  constexpr uint32_t GQ = 32 * NB;
  intx4 Bq[NB][2][2];
  const uint4* const cn_src = tab8 + HS_J8_CONST_AT;
  const uint32_t low_half = up ? 0u : 0xffffffffu;
#define HS_LOAD_MEMBERS(D0)                                                              \
  {                                                                                      \
    const int64_t off_ = (int64_t)(((uint64_t)(D0).y << 32) | (uint64_t)(D0).x);         \
    const uint4* src_ = (up ? rec_base : packed_base) + off_;                            \
    uint32_t idx_ = (D0).w * 128u + (uint32_t)n;                                         \
    if (q == 3) {                                                                        \
      src_ = cn_src - ((D0).z - 1u);                                                     \
      idx_ = 0x7fffff00u;                                                                \
    }                                                                                    \
    _Pragma("unroll") for (int t = 0; t < RT; ++t)                                       \
      mk[t] = src_[min(idx_ + 16 * t, (D0).z - 1)];                                      \
  }
  const uint32_t skew = ((blockIdx.x * 4u + (uint32_t)wave) * 40503u) & 0xffffu;
#define HS_N_GROUPS(D1) (((D1).z - (D1).y + GQ - 1u) / GQ)
#define HS_FIRST_Q(D1) ((D1).y + GQ * ((skew * HS_N_GROUPS(D1)) >> 16))
#define HS_LOAD_GROUP(ROW, Q0, QEND)                                                             \
  _Pragma("unroll") for (int u = 0; u < NB; ++u) {                                               \
    const uint32_t qu_ = (Q0) + 32u * u < (QEND) ? (Q0) + 32u * u : (Q0);                        \
    load_btile_x(Bq[u], c8t, (ROW) + qu_, min(32u, (QEND) - qu_), lane);                         \
  }
  HS_LOAD_MEMBERS(d0)
  {
    const uint32_t q0 = HS_FIRST_Q(d1);
    HS_LOAD_GROUP(d1.x, q0, d1.z)
  }
  while (true) {
    const uint32_t M = d0.z, mt = d0.w;
    const uint32_t qoff = d1.x, q_begin = d1.y, q_end = d1.z, mstart = d1.w;
    const uint32_t wbase = mt * 128u;
    const bool has_next = next_item < n_items;
    HS_ADVANCE_PF()  // pf_item = the item after next
    HS_T(0)
    uint4 nnd0 = nd0, nnd1 = nd1;
    if (pf_item < n_items) {
      nnd0 = uniform4(desc[2 * (uint64_t)pf_item]);
      nnd1 = uniform4(desc[2 * (uint64_t)pf_item + 1]);
    }
    // ---- A operands of the item's 128 members
    HS_TD(6, mk[0].x ^ mk[7].w)   // wait for the members
    intx4 A[RT][2];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      // Every lane needs the member's packed word shifted down by 20 q bits (positions 4 q + m of
      // k-step 0 at bits 5 m, 16 + 4 q + m of k-step 1 at bits 80 + 5 m).  The upper half-wave gets
      // it from the lower one already moved down by one dword: v_permlane32_swap(a, b) exchanges the
      // upper half of a with the lower half of b, so swap(x, y) leaves {own x | the lower lane's y},
      // and so on -- three swaps and one mask instead of four swaps and four selects.  The upper
      // half's own registers (record / constants) are not touched.
      const uint32_t x0 = __builtin_amdgcn_permlane32_swap(mk[t].x, mk[t].y, false, false)[0];
      const uint32_t y0 = __builtin_amdgcn_permlane32_swap(mk[t].y, mk[t].z, false, false)[0];
      const uint32_t z0 = __builtin_amdgcn_permlane32_swap(mk[t].z, mk[t].w, false, false)[0];
      const uint32_t w0 = mk[t].w & low_half;
      const uint32_t bs = (20u * (uint32_t)q) & 31u;  // 0, 20, 8, 28
      const uint32_t x = __funnelshift_r(x0, y0, bs), z = __funnelshift_r(z0, w0, bs), w = w0 >> bs;
      // ten-bit fields at bits 0, 10 (k-step 0) and 80, 90 (k-step 1: bits 16..25 / 26..35 of z:w)
      const uint2 p0 = sPair[x & 1023u], p1 = sPair[(x >> 10) & 1023u];
      const uint2 p2 = sPair[(z >> 16) & 1023u], p3 = sPair[__funnelshift_r(z, w, 26) & 1023u];
      A[t][0] = intx4{(int)p0.x, (int)p0.y, (int)p1.x, (int)p1.y};
      const intx4 lk = intx4{(int)p2.x, (int)p2.y, (int)p3.x, (int)p3.y};
      const intx4 own = intx4{(int)mk[t].x, (int)mk[t].y, (int)mk[t].z, (int)mk[t].w};
      A[t][1] = up ? own : lk;  // q = 2: the member's record, q = 3: the constants it loaded
    }
    HS_TD(1, __builtin_amdgcn_readfirstlane(A[0][0][0] ^ A[7][1][3] ^ A[3][1][0]))
    HS_LOAD_MEMBERS(nd0)
    HS_T(2)
    intx4 accX[4][2], accY[4][2];
    // Y has no previous query tile at the item's first one: its sign test runs on whatever the
    // registers hold (no 32 moves per item to preset them) and y_live says not to believe it
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c) accY[t][c] = __builtin_nondeterministic_value(accY[t][c]);
    bool y_live = false;
    uint32_t prev_qc = q_begin;
    const uint32_t n_groups = HS_N_GROUPS(d1);
    uint32_t qc0 = HS_FIRST_Q(d1);
    auto do_group = [&](uint32_t gi) {
      uint32_t nrow = qoff, nq0 = qc0 + GQ, nqend = q_end;
      if (nq0 >= q_end) nq0 = q_begin;
      if (gi + 1 == n_groups) {
        nrow = nd1.x;
        nq0 = HS_FIRST_Q(nd1);
        nqend = nd1.z;
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const uint32_t qc = qc0 + 32u * (uint32_t)u;
        intx4 (&B)[2][2] = Bq[u];
        if (u == 0 || qc < q_end) {
          // ---- phase 1: X <- row tiles 0..3 x B, beside the sign test of Y (previous query tile)
          const uint32_t sY = and_tree_x(accY);
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int c = 0; c < 2; ++c)
                accX[t][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t][s], B[s][c],
                                                                   s ? accX[t][c] : intx4{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          }
          if (y_live && __ballot((int)sY >= 0))
            emit_survivors_x(accY, 4, prev_qc, qoff, q_end, wbase, M, mstart, lane, res_base, res_used,
                             prov_count, prov_cap, prov);
          y_live = true;
          // ---- phase 2: Y <- row tiles 4..7 x B, beside the sign test of X
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int c = 0; c < 2; ++c)
                accY[t][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[4 + t][s], B[s][c],
                                                                   s ? accY[t][c] : intx4{0, 0, 0, 0}, 0, 0, 0);
          const uint32_t sX = and_tree_x(accX);
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          }
          if (__ballot((int)sX >= 0))
            emit_survivors_x(accX, 0, qc, qoff, q_end, wbase, M, mstart, lane, res_base, res_used,
                             prov_count, prov_cap, prov);
          prev_qc = qc;
        }
        const uint32_t nb = nq0 + 32u * (uint32_t)u < nqend ? nq0 + 32u * (uint32_t)u : nq0;
        load_btile_x(B, c8t, nrow + nb, min(32u, nqend - nb), lane);
      }
      qc0 = nq0;
    };
    do_group(0);
    HS_T(3)
    for (uint32_t gi = 1; gi < n_groups; ++gi) do_group(gi);
    HS_T(4)
    {  // the item's last Y group
      const uint32_t sY = and_tree_x(accY);
      if (__ballot((int)sY >= 0))
        emit_survivors_x(accY, 4, prev_qc, qoff, q_end, wbase, M, mstart, lane, res_base, res_used,
                         prov_count, prov_cap, prov);
    }
    HS_T(5)
    if (!has_next) break;
    item = next_item;
    next_item = pf_item;
    d0 = nd0;
    d1 = nd1;
    nd0 = nnd0;
    nd1 = nnd1;
  }
#undef HS_ADVANCE_PF
#undef HS_TAKE_CHUNK
#undef HS_LOAD_GROUP
#undef HS_N_GROUPS
#undef HS_LOAD_MEMBERS
#undef HS_FIRST_Q
  close_reservation(prov, res_base, res_used, prov_cap, lane);
#ifdef HS_JOIN_TIMING
  if (lane == 0) {
    for (int i = 0; i < 7; ++i) atomicAdd(&g_join8_timing[i], (unsigned long long)tacc[i]);
    atomicMax(&g_join8_timing[7], (unsigned long long)(tlast - tstart));
  }
#endif
}


// ---- hs_join8xw_kernel: the 16x16x64 form for WIDE rows of 192 bytes (k <= 20, all 8 coordinate columns) ------
// Same operands, filter value, work items (64 members) and survivor list as hs_join8w_kernel<2, 6, true>; the
// structure of hs_join8x_kernel: lane (n, q) holds bytes 64 s + 16 q .. + 15 of k-step s = the two positions
// 8 s + 2 q, 8 s + 2 q + 1 (8 coordinate bytes each) for s = 0, 1 and for s = 2, q < 2; in k-step 2 lane
// quarter q = 2 carries the member's record (bytes 160..175 of the row) and q = 3 the constant factors of the
// gamma slots (176..191).  A lane's two positions are ONE ten-bit field of the packed word (bit 40 s + 10 q),
// and the table holds the 16 bytes of both residues per entry (r1 << 5 | r0: 16 KB of LDS): one 16-byte
// lookup per k-step, lane and row tile IS the operand register.  EIGHT row tiles per item (128 members: with
// 64 the query tiles a wave streams from L2 meet half as many members, and at 192 bytes per query row that
// traffic -- 15 TB/s asked of the L2 at k = 15 -- was the kernel's limit, not the matrix pipe), two
// accumulator sets of two row tiles that the four row-tile pairs of a query tile take in turn, each pair's
// sign test in the gaps of the next pair's MFMAs.
template <int NS>
__device__ __forceinline__ void load_btile_xn(intx4 (&B)[NS][2], const uint4* __restrict__ c8t, uint32_t row0,
                                              uint32_t nr, int lane) {
  // row = 4 NS pieces of 16 bytes; piece 4 s + q of row 16 c + n of a tile with nr rows sits at (4 s + q) nr + row
  const char* t0 = reinterpret_cast<const char*>(c8t + (uint64_t)row0 * (4u * NS));
  const uint32_t n = (uint32_t)lane & 15u, q = (uint32_t)lane >> 4;
  const uint32_t qn = q * nr;
  const uint32_t o0 = (qn + min(n, nr - 1u)) * 16u, o1 = (qn + min(16u + n, nr - 1u)) * 16u;
#pragma unroll
  for (int s2 = 0; s2 < NS; ++s2) {
    const char* ts = t0 + (uint64_t)nr * 64u * (uint32_t)s2;
    const uint4 v0 = *reinterpret_cast<const uint4*>(ts + o0), v1 = *reinterpret_cast<const uint4*>(ts + o1);
    B[s2][0] = intx4{(int)v0.x, (int)v0.y, (int)v0.z, (int)v0.w};
    B[s2][1] = intx4{(int)v1.x, (int)v1.y, (int)v1.z, (int)v1.w};
  }
}

// Survivors in hs_join8xw_kernel.  Short k-mers at a loose radius pass 0.1 % of the pairs through the
// filter (k = 15 at the C2 sizes: five survivors per 128 x 32 query tile), so nearly every query tile has
// some.  Emitting them per 16 x 16 tile with ballot loops (emit_survivors_x2: 22.8 ms against 14.3 for the
// kernel without them) or one by one with scalar code (22.4: ~ 230 cycles of dependent scalar instructions
// per survivor) both cost as much as the MFMAs; even a branch per row-tile pair that only saved the pair's
// sign bits when it had a survivor cost 4 ms (18.2 against 14.3).  Here EVERY pair's 16 sign bits per lane
// go into a 64-bit per-lane mask (bit 16 pair + 8 t + 4 c + i) with straight-line code in the MFMA gaps, and
// the query tile's survivors are written out ONCE, after its fourth pair: one round per "r-th survivor
// of a lane".
// bit 8 t + 4 c + i = 1 where acc[t][c][i] >= 0 (a survivor): 16 v_alignbit, branch-free -- they take the
// place of the sign AND-tree in the gaps of the next pair's MFMAs
__device__ __forceinline__ uint32_t sign_mask16(const intx4 (&acc)[2][2]) {
  uint32_t neg = 0;
#pragma unroll
  for (int t = 1; t >= 0; --t)
#pragma unroll
    for (int c = 1; c >= 0; --c)
#pragma unroll
      for (int i = 3; i >= 0; --i) neg = __builtin_amdgcn_alignbit(neg, (uint32_t)acc[t][c][i], 31);
  return ~neg & 0xffffu;
}



// hs_join8xw_kernel's survivors go through a per-wave LDS buffer of this many entries and take exactly
// as many slots of the list when it fills.  A store to the list inside the query-tile loop -- in whatever
// form: its trip count is unknown -- makes the compiler drain vmcnt(0) before the next use of a prefetched
// query tile (stores count with the loads on this part), once per query tile in a hit-heavy batch: that
// drain, not the emission's instructions, was the 7 ms every form of it cost.
constexpr uint32_t JRESW = 256;
// the wave's buffered survivors to the list: exactly as many slots as there are entries
__device__ __forceinline__ void drain_survivors_xw(const uint2* sbuf, uint32_t& n_buf, int lane,
                                                   uint32_t* __restrict__ prov_count, uint32_t prov_cap,
                                                   uint2* __restrict__ prov) {
  if (!n_buf) return;
  uint32_t base = 0;
  if (lane == 0) base = hs_reserve_survivors(prov_count, n_buf);
  base = __builtin_amdgcn_readfirstlane(base);
  __builtin_amdgcn_wave_barrier();
  for (uint32_t i = (uint32_t)lane; i < n_buf; i += 64u)
    if (base + i < prov_cap) prov[base + i] = sbuf[i];
  __builtin_amdgcn_wave_barrier();
  n_buf = 0;
}

__device__ __forceinline__ void flush_stash_xw(uint64_t& st, uint32_t qc, uint32_t qoff,
                                               uint32_t wbase, uint32_t M, uint32_t mstart, int lane,
                                               uint2* sbuf, uint32_t& n_buf,
                                               uint32_t* __restrict__ prov_count, uint32_t prov_cap,
                                               uint2* __restrict__ prov) {
  const uint32_t n = (uint32_t)lane & 15u, q = (uint32_t)lane >> 4;
  // (one 64-bit value per lane, not a pair of words picked by a branch: that form -- a store through a
  // selected pointer -- kept the two words in scratch memory)
  while (__ballot(st != 0)) {
    uint32_t idx = 0, col = 0;
    bool pass = false;
    if (st) {
      const uint32_t b = (uint32_t)__builtin_ctzll(st);  // the lane's lowest set bit: 16 pair + 8 t + 4 c + i
      st &= st - 1ull;
      idx = wbase + 16u * (b >> 3) + 4u * q + (b & 3u);  // row tile 2 pair + t = b >> 3
      col = qc + 16u * ((b >> 2) & 1u) + n;
      pass = idx < M;
    }
    const unsigned long long m = __ballot(pass);
    if (m) {
      const uint32_t cnt = (uint32_t)__popcll(m);
      if (n_buf + cnt > JRESW) drain_survivors_xw(sbuf, n_buf, lane, prov_count, prov_cap, prov);
      if (pass)
        sbuf[n_buf + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] =
            make_uint2(HS_PROV_INDIRECT | (qoff + col), mstart + idx);
      n_buf += cnt;
    }
  }
}

__global__ __launch_bounds__(256, 2) void hs_join8xw_kernel(
    const uint4* __restrict__ desc, uint32_t n_items, const uint4* __restrict__ packed_base,
    const uint4* __restrict__ rec_base, const uint4* __restrict__ c8t,
    const uint4* __restrict__ tabW, uint32_t* __restrict__ prov_count, uint32_t prov_cap,
    uint2* __restrict__ prov, uint32_t* __restrict__ item_counter, uint32_t G,
    const uint32_t* __restrict__ n_items_dev) {
  if (n_items_dev) n_items = min(n_items, __builtin_amdgcn_readfirstlane(*n_items_dev));
  constexpr int RT = 8;  // row tiles of 16 members per item: 128 members meet every query tile a wave fetches
  constexpr int NS = 3;  // k-steps of 64 bytes
  // both residues of a ten-bit field: entry (r1 << 5 | r0) = {x^(r0) columns 0..3, 4..7, x^(r1) 0..3, 4..7}
  __shared__ uint4 sPairW[1024];
  __shared__ uint2 sSurv[4][JRESW];  // per wave: survivors on their way to the list
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4, up = lane >> 5;
  uint2* const sbuf = sSurv[wave];
  uint32_t n_buf = 0;
  for (int e = tid; e < 1024; e += 256)
    sPairW[e] = make_uint4(tabW[e & 31].x, tabW[e & 31].y, tabW[e >> 5].x, tabW[e >> 5].y);
  __syncthreads();  // the only one: the table is read-only from here on
  const uint32_t first_dynamic = gridDim.x * 4u * G;
  uint32_t item = (blockIdx.x * 4u + (uint32_t)wave) * G;
  if (item >= n_items) return;
  uint32_t next_chunk_v = 0;
  if (lane == 0) next_chunk_v = atomicAdd(item_counter, G);
  uint32_t pf_item = item, pf_chunk_end = item + G;
#define HS_ADVANCE_PF()                                                                  \
  {                                                                                      \
    ++pf_item;                                                                           \
    if (pf_item == pf_chunk_end) {                                                       \
      pf_item = first_dynamic + __builtin_amdgcn_readfirstlane(next_chunk_v);            \
      pf_chunk_end = pf_item + G;                                                        \
      if (lane == 0 && pf_item < n_items) next_chunk_v = atomicAdd(item_counter, G);     \
    }                                                                                    \
  }
  uint4 d0 = uniform4(desc[2 * (uint64_t)item]), d1 = uniform4(desc[2 * (uint64_t)item + 1]);
  HS_ADVANCE_PF()
  uint32_t next_item = pf_item;
  uint4 nd0 = d0, nd1 = d1;
  if (next_item < n_items) {
    nd0 = uniform4(desc[2 * (uint64_t)next_item]);
    nd1 = uniform4(desc[2 * (uint64_t)next_item + 1]);
  }
  // lanes 0..31: packed member 16 t + n (both lane quarters the same word); lanes 32..63: its record
  uint4 mk[RT];
  constexpr int NB = 2;          // B tiles in flight per wave: the one in use + one prefetched
  constexpr uint32_t GQ = 32 * NB;
  intx4 Bq[NB][NS][2];
#define HS_LOAD_MEMBERS(D0)                                                              \
  {                                                                                      \
    const int64_t off_ = (int64_t)(((uint64_t)(D0).y << 32) | (uint64_t)(D0).x);         \
    const uint4* src_ = (up ? rec_base : packed_base) + off_;                            \
    const uint32_t idx_ = (D0).w * (16u * RT) + (uint32_t)n;                             \
    _Pragma("unroll") for (int t = 0; t < RT; ++t)                                       \
      mk[t] = src_[min(idx_ + 16 * t, (D0).z - 1)];                                      \
  }
  const uint32_t skew = ((blockIdx.x * 4u + (uint32_t)wave) * 40503u) & 0xffffu;
#define HS_N_GROUPS(D1) (((D1).z - (D1).y + GQ - 1u) / GQ)
#define HS_FIRST_Q(D1) ((D1).y + GQ * ((skew * HS_N_GROUPS(D1)) >> 16))
#define HS_LOAD_GROUP(ROW, Q0, QEND)                                                             \
  _Pragma("unroll") for (int u = 0; u < NB; ++u) {                                               \
    const uint32_t qu_ = (Q0) + 32u * u < (QEND) ? (Q0) + 32u * u : (Q0);                        \
    load_btile_xn<NS>(Bq[u], c8t, (ROW) + qu_, min(32u, (QEND) - qu_), lane);                    \
  }
  HS_LOAD_MEMBERS(d0)
  {
    const uint32_t q0 = HS_FIRST_Q(d1);
    HS_LOAD_GROUP(d1.x, q0, d1.z)
  }
  const uint32_t bs = 10u * (uint32_t)q;  // a lane's fields start 10 q bits into each 40-bit stretch
  while (true) {
    const uint32_t M = d0.z, mt = d0.w;
    const uint32_t qoff = d1.x, q_begin = d1.y, q_end = d1.z, mstart = d1.w;
    const uint32_t wbase = mt * (16u * RT);
    const bool has_next = next_item < n_items;
    HS_ADVANCE_PF()  // pf_item = the item after next
    uint4 nnd0 = nd0, nnd1 = nd1;
    if (pf_item < n_items) {
      nnd0 = uniform4(desc[2 * (uint64_t)pf_item]);
      nnd1 = uniform4(desc[2 * (uint64_t)pf_item + 1]);
    }
    // ---- A operands of the item's 128 members
    intx4 A[RT][NS];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      // the upper half-wave takes the packed word's first three dwords from the lower one (v_permlane32_swap
      // (a, b) exchanges the upper half of a with the lower half of b); its own registers hold the record
      const uint32_t x0 = __builtin_amdgcn_permlane32_swap(mk[t].x, mk[t].x, false, false)[0];
      const uint32_t y0 = __builtin_amdgcn_permlane32_swap(mk[t].y, mk[t].y, false, false)[0];
      const uint32_t z0 = __builtin_amdgcn_permlane32_swap(mk[t].z, mk[t].z, false, false)[0];
      const uint32_t x = __funnelshift_r(x0, y0, bs), y = __funnelshift_r(y0, z0, bs);
      const uint32_t z = __funnelshift_r(z0, mk[t].w, bs);  // (used by the lower half only: its own w)
      const uint4 p0 = sPairW[x & 1023u], p1 = sPairW[(y >> 8) & 1023u], p2 = sPairW[(z >> 16) & 1023u];
      A[t][0] = intx4{(int)p0.x, (int)p0.y, (int)p0.z, (int)p0.w};
      A[t][1] = intx4{(int)p1.x, (int)p1.y, (int)p1.z, (int)p1.w};
      const intx4 lk = intx4{(int)p2.x, (int)p2.y, (int)p2.z, (int)p2.w};
      const intx4 rec = intx4{(int)mk[t].x, (int)mk[t].y, (int)mk[t].z, (int)mk[t].w};
      const intx4 cn = intx4{0x7f7f0000, 0x7f7f7f7f, 0x7f7f7f7f, 0x017f7f7f};  // (build_afrags8_wide, h = 1)
      A[t][2] = q == 3 ? cn : q == 2 ? rec : lk;
    }
    // Two accumulator sets of two row tiles each; a query tile runs four phases (row-tile pairs 0..3)
    // that alternate between them, each phase's 12 MFMAs beside the sign test of the phase before (the
    // first phase of a query tile: of the previous tile's last phase).
    intx4 acc[2][2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[1][t][c] = __builtin_nondeterministic_value(acc[1][t][c]);
    bool y_live = false;  // (nothing stands in acc[1] at the item's first phase)
    uint64_t st = 0;  // per lane: the sign bits of the current query tile's four pairs
    uint32_t prev_qc = q_begin;
    const uint32_t n_groups = HS_N_GROUPS(d1);
    uint32_t qc0 = HS_FIRST_Q(d1);
    auto do_group = [&](uint32_t gi) {
      uint32_t nrow = qoff, nq0 = qc0 + GQ, nqend = q_end;
      if (nq0 >= q_end) nq0 = q_begin;
      if (gi + 1 == n_groups) {
        nrow = nd1.x;
        nq0 = HS_FIRST_Q(nd1);
        nqend = nd1.z;
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const uint32_t qc = qc0 + 32u * (uint32_t)u;
        intx4 (&B)[NS][2] = Bq[u];
        if (u == 0 || qc < q_end) {
#pragma unroll
          for (int ph = 0; ph < 4; ++ph) {
            intx4 (&cur)[2][2] = acc[ph & 1];
            intx4 (&old)[2][2] = acc[(ph & 1) ^ 1];
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
              for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                  cur[t][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2 * ph + t][s2], B[s2][c],
                                                                    s2 ? cur[t][c] : intx4{0, 0, 0, 0}, 0, 0, 0);
            // the pair before: pair ph - 1 of this query tile, or pair 3 of the previous one (nothing at the
            // item's very first pair) -- its sign bits into the tile's mask
            const uint32_t sm = sign_mask16(old);
            if (ph == 0) st |= y_live ? (uint64_t)sm << 48 : 0ull;
            if (ph > 0) st |= (uint64_t)sm << (16 * (ph - 1));
#pragma unroll
            for (int g = 0; g < 12; ++g) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
            if (ph == 0) {  // the previous query tile is complete: its survivors, if any, go out
              const uint64_t bad0 = prev_qc + (uint32_t)n < q_end ? 0ull : 0x0f0f0f0f0f0f0f0full;
              const uint64_t bad1 = prev_qc + 16u + (uint32_t)n < q_end ? 0ull : 0xf0f0f0f0f0f0f0f0ull;
              st &= ~(bad0 | bad1);
              if (__ballot(st != 0))
                flush_stash_xw(st, prev_qc, qoff, wbase, M, mstart, lane, sbuf, n_buf, prov_count, prov_cap, prov);
            }
          }
          y_live = true;
          prev_qc = qc;
        }
        const uint32_t nb = nq0 + 32u * (uint32_t)u < nqend ? nq0 + 32u * (uint32_t)u : nq0;
        load_btile_xn<NS>(B, c8t, nrow + nb, min(32u, nqend - nb), lane);
      }
      qc0 = nq0;
    };
    do_group(0);
    for (uint32_t gi = 1; gi < n_groups; ++gi) do_group(gi);
    // The next item's members are requested HERE, not beside the operand build as in hs_join8x_kernel: eight
    // more live 16-byte registers through the query-tile loop were three spilled registers, and scratch
    // traffic in that loop -- it counts with the query-tile loads -- cost what the emission seemed to
    // cost (22 ms against 14 without it, whatever its form).  An item of these rows has tens of query tiles.
    HS_LOAD_MEMBERS(nd0)
    {  // the item's last pair
      st |= (uint64_t)sign_mask16(acc[1]) << 48;
      const uint64_t bad0 = prev_qc + (uint32_t)n < q_end ? 0ull : 0x0f0f0f0f0f0f0f0full;
      const uint64_t bad1 = prev_qc + 16u + (uint32_t)n < q_end ? 0ull : 0xf0f0f0f0f0f0f0f0ull;
      st &= ~(bad0 | bad1);
      if (__ballot(st != 0))
        flush_stash_xw(st, prev_qc, qoff, wbase, M, mstart, lane, sbuf, n_buf, prov_count, prov_cap, prov);
    }
    if (!has_next) break;
    item = next_item;
    next_item = pf_item;
    d0 = nd0;
    d1 = nd1;
    nd0 = nnd0;
    nd1 = nnd1;
  }
#undef HS_ADVANCE_PF
#undef HS_LOAD_GROUP
#undef HS_N_GROUPS
#undef HS_LOAD_MEMBERS
#undef HS_FIRST_Q
  drain_survivors_xw(sbuf, n_buf, lane, prov_count, prov_cap, prov);
}

// ------------------------------------------------------------------ join, query-resident form
// Segments probed by FEW queries of the batch (<= HS_JR_MAXQ = 64: configs[2]'s shape at its swept W
// has ~ 28 per work item, 1.7e7 work items per step).  hs_join8x_kernel treats every work item alike:
// descriptor, member operands, then the item's query tiles streamed from L2 -- three tiles fetched per
// item whether they exist or not, with their address arithmetic; at one or two query tiles per item
// that fixed cost is the kernel (measured: 449 vector instructions and 51 MFMAs per item, the vector
// pipe busier than the matrix pipe).  Here the QUERIES are the stationary operand: a segment's <= 64
// query rows are loaded into registers once, when the wave enters the segment, and the wave then only
// streams member tiles through them -- consecutive work items of the item list are consecutive
// 128-member tiles of the same bucket.  Per item what remains is the member operand build, the MFMAs of
// the 16-query column tiles that EXIST (1..4: no padding to 32 columns, no tile of a neighbour), the sign
// test, and eight loads for the item two ahead (two register sets, so that a member tile has two item
// times to arrive from HBM).  Same work items, descriptors, operands, filter value and survivor list as
// hs_join8x_kernel; the item list's tail [range[0], range[1]) is this kernel's share.
//
// Loads and waits.  The matrix loop waits on vector-memory loads by count, in issue order.  The query
// tiles are (re)loaded inside a branch -- segment changed -- and the compiler would charge the code
// after the branch with a full drain; so the branch itself consumes the tiles it loaded (an empty asm
// that reads the registers): the drain sits inside the branch, once per segment, and the common path
// only ever waits for the member tile it is about to use.
// Query tile of hs_join8r_kernel: the rows are the same (hs_gather_c8t_kernel), the K layout is the
// member side's -- lane row 0, 1, 3 carries positions 8 j .. 8 j + 3 in k-step 0 and 8 j + 4 .. 8 j + 7 in
// k-step 1 (j = 0, 1, 2), row 2 the record, then the constants -- so lane (n, row) takes the 16-byte
// pieces {0, 2, 6, 4}[row] and {1, 3, 7, 5}[row] of query row 16 c + n (piece g of row j of a tile with
// nr rows sits at g nr + j).
__device__ __forceinline__ void load_btile_r(intx4 (&B)[2][2], const uint4* __restrict__ c8t, uint32_t row0,
                                             uint32_t nr, int lane) {
  const char* t0 = reinterpret_cast<const char*>(c8t + (uint64_t)row0 * 8);
  const uint32_t n = (uint32_t)lane & 15u, row = (uint32_t)lane >> 4;
  const uint32_t g0 = row == 0 ? 0u : row == 1 ? 2u : row == 2 ? 6u : 4u;
  const uint32_t r0 = min(n, nr - 1u), r1 = min(16u + n, nr - 1u);
  const uint32_t o00 = (g0 * nr + r0) * 16u, o01 = (g0 * nr + r1) * 16u;
  const uint32_t o10 = ((g0 + 1u) * nr + r0) * 16u, o11 = ((g0 + 1u) * nr + r1) * 16u;
  const uint4 v00 = *reinterpret_cast<const uint4*>(t0 + o00), v01 = *reinterpret_cast<const uint4*>(t0 + o01);
  const uint4 v10 = *reinterpret_cast<const uint4*>(t0 + o10), v11 = *reinterpret_cast<const uint4*>(t0 + o11);
  // row 2 (the record's position | the gamma slots): dword 0 of the two k-steps change places, as the member
  // side holds them (join8r_layout)
  const bool sw = row == 2u;
  B[0][0] = intx4{(int)(sw ? v10.x : v00.x), (int)v00.y, (int)v00.z, (int)v00.w};
  B[0][1] = intx4{(int)(sw ? v11.x : v01.x), (int)v01.y, (int)v01.z, (int)v01.w};
  B[1][0] = intx4{(int)(sw ? v00.x : v10.x), (int)v10.y, (int)v10.z, (int)v10.w};
  B[1][1] = intx4{(int)(sw ? v01.x : v11.x), (int)v11.y, (int)v11.z, (int)v11.w};
}

// hs_join8r_kernel works on HALF items: four row tiles (64 members) at a time -- the whole item's
// operands would not fit beside two member sets in flight -- in two steps that the kernel software-
// pipelines across halves and items:
//   join8r_lookup   the half's table lookups are ISSUED (addresses from the lanes' loaded member words) ...
//   join8r_half     ... and one half-step later, when they have long arrived, the operand is completed
//                   (row 2 takes its record) and meets the NCT 16-query column tiles of the resident query
//                   rows: per 32-query tile the MFMAs, then their sign test.
// Lanes of rows 0, 1, 3 hold 16 bytes of the member's packed word from the dword their eight positions
// start in (bit sh of it): four ten-bit fields = four pair lookups, two per k-step.  Row 2 holds the
// member's record in FOUR bytes (hs_gather_rec8_kernel's out_rho: the kernel is bound by the bytes it reads
// per member, 16 + 4 instead of 16 + 16): the word carries the byte offset of the table entry that holds
// the record's twelve digit bytes, looked up in the two slots the other rows use for their first two pairs;
// its k-step 1 are the two table entries with the constant factors (m2 = 0 cancels its field bits, rc2 / rc3
// address them).  No lane needs another lane's load: the query rows are permuted to this layout instead
// (load_btile_r).
// A group (half x tile) with a survivor -- a few per cent of them -- leaves its 32 sign bits per lane
// (bit 8 t + 4 c + i = accumulator i of row tile t, column tile c) in the wave's LDS slot of that group
// and its number in the returned mask; the item's survivors are written out ONCE, after the item, by the
// caller (the survivor code inlined at every group made the kernel 140 KB of instructions: 2.2 x the
// instruction cache).
// per-lane constants of the operand build (rows 0, 1, 3 | row 2)
struct J8rLane {
  uint32_t sh;      // bit of the loaded words the lane's fields start at: 8 j | 0
  uint32_t o0, o1;  // bit offsets of the first two fields: 0, 10 | 0, 0 (the record word carries |q| there)
  uint32_t w01;     // ... and their width: 10 | 11
  uint32_t sl;      // field -> byte offset of its table entry: shift 6 (64-byte entries) | 3 (8-byte entries)
  uint32_t b0, b1;  // the lane's copy of the pair table | the two digit tables
  uint32_t w3;      // k-step 1: width of its second field: 10 | 0 (no field: row 2 reads the constant factors)
  uint32_t b2, b3;  // the lane's copy again | the table of single residues (field = the record's residue), the
                    // second half of the constant factors
  uint32_t mrem;    // 0 | 0x7f
  bool row2;
};
// LDS of hs_join8r_kernel, for an alphabet of A <= HS_JR_MAX_ALPHABET residues (byte offsets):
//   [0, 2048 A)      x^ of TWO consecutive residues per lookup: entry r1 << 5 | r0 = {x^(r0), x^(r1)}, 64 bytes
//                    each = EIGHT copies side by side -- lane l reads copy l & 7.  The 16 lanes of one pass
//                    of an 8-byte read then meet, copy by copy, in 4 bank groups two at a time (one copy:
//                    72 % of the LDS cycles were conflicts; four copies: as many conflict cycles as data
//                    cycles, and the LDS pipe was the busiest unit of the kernel at ~ 70 %)
//   digit tables     the record digits of every |q| (digit j = clamp(|q| - 127 j, 0, 127), 1398 entries) as
//                    TWO arrays of 8-byte entries -- {0, digits 0..3} and {digits 4..7, digits 8..10} -- so
//                    that the 16 lanes of row 2, whose |q| are neighbours, spread over 32 bank pairs and not
//                    over the 16 that 16-byte entries gave them
//   x^ of ONE residue: entry r (stride = the pair table's) = {x^(r), second dword of the constant factors},
//                    33 entries, entry 32 = {0, ..}: no residue.  Row 2 reads it in the slot of k-step 1's first
//                    pair, so its operand dwords are {factor 0, digits ..} | {x^, factors 1..3}: dword 0 of the
//                    two k-steps swapped against the layout of the query rows, which load_btile_r swaps back.
//                    (Until r04 a lookup of its own -- three vector instructions and an LDS read per row tile
//                    for ALL lanes -- and a select put x^ into k-step 0.)
//   the constant factors of the gamma slots (16 bytes)
//   per wave: the sign masks of an item's groups that had a survivor, [4][8][64] dwords
struct J8rLayout {
  uint32_t dig_at, dig2_at, tab1_at, const_at, mask_at, total;
};
__host__ __device__ inline J8rLayout join8r_layout(int alphabet) {
  J8rLayout l;
  l.dig_at = 2048u * (uint32_t)alphabet;
  l.dig2_at = l.dig_at + 1398u * 8u;
  l.tab1_at = (l.dig2_at + 1398u * 8u + 255u) & ~255u;
  l.const_at = l.tab1_at + 33u * 64u;
  l.mask_at = l.const_at + 256u;
  l.total = l.mask_at + 4u * 8u * 64u * 4u;
  return l;
}

__device__ __forceinline__ void join8r_lookup(intx4 (&A)[4][2], const uint2 (&MK)[8], int half, const char* sL,
                                              const J8rLane& c) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint2 mk = MK[4 * half + t];  // dwords j, j + 1 of the packed word (rows 0, 1, 3) | the record (row 2)
    // the lane's stream: bits 0..31, and 32 .. 63 - sh of which 32..39 are needed
    const uint32_t x0 = __builtin_amdgcn_alignbit(mk.y, mk.x, c.sh), x1 = mk.y >> c.sh;
    const uint32_t a0 = (__builtin_amdgcn_ubfe(x0, c.o0, c.w01) << c.sl) + c.b0;
    const uint32_t a1 = (__builtin_amdgcn_ubfe(x0, c.o1, c.w01) << c.sl) + c.b1;
    const uint32_t a2 = (__builtin_amdgcn_ubfe(x0, 20u, 10u) << 6) + c.b2;
    const uint32_t a3 = (__builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(x1, x0, 30), 0u, c.w3) << 6) + c.b3;
    const uint2 p0 = *reinterpret_cast<const uint2*>(sL + a0), p1 = *reinterpret_cast<const uint2*>(sL + a1);
    const uint2 p2 = *reinterpret_cast<const uint2*>(sL + a2), p3 = *reinterpret_cast<const uint2*>(sL + a3);
    A[t][0] = intx4{(int)p0.x, (int)p0.y, (int)p1.x, (int)p1.y};
    A[t][1] = intx4{(int)p2.x, (int)p2.y, (int)p3.x, (int)p3.y};
  }
}

// the lookups have arrived (this is where the wave waits for them): row 2's k-step 0 is the digit
// pattern of |q| it looked up (first dword: the first constant factor); it becomes the member's record
// with rho mod 127 in its last byte, and the digits negated should q be negative
__device__ __forceinline__ void join8r_finish(intx4 (&A)[4][2], const uint2 (&MK)[8], int half, const J8rLane& c) {
  uint32_t any_neg = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint32_t w = MK[4 * half + t].x;
    A[t][0][3] = (int)((((w >> 12) & c.mrem) << 24) | (uint32_t)A[t][0][3]);
    any_neg |= w;
  }
  if (__ballot(c.row2 && (any_neg & 0x800u))) {  // a negative rho: a k-mer of very small norm (rare)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t w = MK[4 * half + t].x;
      if (c.row2 && (w & 0x800u)) {
        // -b per byte for digits b in 0..127: ((~v & 0x7f7f7f7f) + 0x01010101) ^ 0x80808080
        const uint32_t y = (uint32_t)A[t][0][1], z = (uint32_t)A[t][0][2], u = (uint32_t)A[t][0][3];
        A[t][0][1] = (int)((((~y) & 0x7f7f7f7fu) + 0x01010101u) ^ 0x80808080u);
        A[t][0][2] = (int)((((~z) & 0x7f7f7f7fu) + 0x01010101u) ^ 0x80808080u);
        const uint32_t un = ((((~u) & 0x007f7f7fu) + 0x00010101u) ^ 0x00808080u) & 0x00ffffffu;
        A[t][0][3] = (int)(un | (u & 0xff000000u));
      }
    }
  }
}

template <int NCT, int HALF>
__device__ __forceinline__ uint32_t join8r_half(const intx4 (&A)[4][2], const intx4 (&Bq)[2][2][2],
                                                uint32_t* __restrict__ smask /* [8][64] + lane */) {
  constexpr int NT = (NCT + 1) / 2;  // 32-query tiles, the last one with one or two 16-query column tiles
  constexpr bool ODD = (NCT & 1) != 0;
  intx4 acc[4][2];
  uint32_t gmask = 0;
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const bool one = ODD && u == NT - 1;  // this tile has a single column tile
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          if (c == 0 || !one)
            acc[t][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t][s2], Bq[u][s2][c],
                                                              s2 ? acc[t][c] : intx4{0, 0, 0, 0}, 0, 0, 0);
    const uint32_t sG = one ? and_tree_x<1>(acc) : and_tree_x<2>(acc);
    if (__ballot((int)sG >= 0)) {
      uint32_t neg = 0;  // bit 8 t + 4 c + i = sign of acc[t][c][i]: shift the sign bits in, last one first
#pragma unroll
      for (int t = 3; t >= 0; --t)
#pragma unroll
        for (int c = 1; c >= 0; --c)
#pragma unroll
          for (int i = 3; i >= 0; --i)
            neg = __builtin_amdgcn_alignbit(neg, (c == 1 && one) ? 0xffffffffu : (uint32_t)acc[t][c][i], 31);
      smask[(HALF * NT + u) * 64] = ~neg;
      gmask |= 1u << (HALF * NT + u);
    }
  }
  return gmask;
}

// The survivors of one item of hs_join8r_kernel from the sign masks its groups left in LDS (see
// join8r_item): group g = half * NT + u covers members wbase + 64 half .. + 63 (row tile t: 16 t + 4 q + i)
// and the queries 32 u + 16 c + n.
__device__ __forceinline__ void join8r_emit(uint32_t gmask, int NT, const uint32_t* __restrict__ smask,
                                            uint32_t qoff, uint32_t nQ, uint32_t wbase, uint32_t M, uint32_t mstart,
                                            int lane, uint32_t& res_base, uint32_t& res_used,
                                            uint32_t* __restrict__ prov_count, uint32_t prov_cap,
                                            uint2* __restrict__ prov) {
  const uint32_t n = (uint32_t)lane & 15u, q = (uint32_t)lane >> 4;
  while (gmask) {
    const int g = __ffs((int)gmask) - 1;
    gmask &= gmask - 1;
    const int half = g / NT, u = g - half * NT;
    uint32_t mask = smask[g * 64];
    while (__ballot(mask != 0)) {
      uint32_t idx = 0, ql = 0;
      bool pass = false;
      if (mask) {
        const int b = __ffs((int)mask) - 1;
        mask &= mask - 1;
        const uint32_t col = 32u * (uint32_t)u + 16u * (uint32_t)((b >> 2) & 1) + n;
        idx = wbase + 64u * (uint32_t)half + 16u * (uint32_t)(b >> 3) + 4u * q + (uint32_t)(b & 3);
        ql = HS_PROV_INDIRECT | (qoff + col);
        pass = idx < M && col < nQ;
      }
      const unsigned long long m = __ballot(pass);
      if (m) {
        const uint32_t cnt = (uint32_t)__popcll(m);
        if (res_used + cnt > JRES) {
          close_reservation(prov, res_base, res_used, prov_cap, lane);
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

__global__ __launch_bounds__(256, 2) void hs_join8r_kernel(
    const uint4* __restrict__ desc, const uint32_t* __restrict__ range, uint32_t desc_cap,
    const uint4* __restrict__ packed_base, const uint32_t* __restrict__ rho_base, const uint4* __restrict__ c8t,
    const uint4* __restrict__ tab8, int alphabet, uint32_t* __restrict__ prov_count,
    uint32_t prov_cap, uint2* __restrict__ prov, uint32_t* __restrict__ item_counter, uint32_t G) {
  const uint32_t first = __builtin_amdgcn_readfirstlane(range[0]);
  const uint32_t n_items = min(__builtin_amdgcn_readfirstlane(range[1]), desc_cap);  // (absolute) end of the list
  extern __shared__ __attribute__((aligned(64))) unsigned char sLds[];  // join8r_layout(alphabet)
  const J8rLayout lay = join8r_layout(alphabet);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, row = lane >> 4;
  {
    uint2* sp = reinterpret_cast<uint2*>(sLds);  // entry f = r1 << 5 | r0, copy c at dword pair 8 f + c
    for (int e = tid; e < alphabet * 32 * 8; e += 256) sp[e] = make_uint2(tab8[(e >> 3) & 31].x, tab8[e >> 8].x);
    uint2* sd = reinterpret_cast<uint2*>(sLds + lay.dig_at);
    uint2* sd2 = reinterpret_cast<uint2*>(sLds + lay.dig2_at);
    const uint4 cf = tab8[HS_J8_CONST_AT];  // the constant factors of the gamma slots
    for (int e = tid; e < 1398; e += 256) {
      uint32_t w[4] = {cf.x, 0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < 11; ++j) {
        const int d = min(127, max(0, e - 127 * j));
        w[(4 + j) >> 2] |= (uint32_t)d << (8 * ((4 + j) & 3));
      }
      sd[e] = make_uint2(w[0], w[1]);
      sd2[e] = make_uint2(w[2], w[3]);
    }
    if (tid < 33) *reinterpret_cast<uint2*>(sLds + lay.tab1_at + 64 * tid) = make_uint2(tid < 32 ? tab8[tid].x : 0u, cf.y);
    if (tid == 0) *reinterpret_cast<uint4*>(sLds + lay.const_at) = cf;
  }
  __syncthreads();  // the only one: the tables are read-only from here on
  uint32_t* const smask = reinterpret_cast<uint32_t*>(sLds + lay.mask_at) + (wave * 8 * 64 + lane);
  const uint32_t first_dynamic = first + gridDim.x * 4u * G;
  uint32_t item = first + (blockIdx.x * 4u + (uint32_t)wave) * G;
  if (item >= n_items) return;
  uint32_t res_base = 0, res_used = JRES;
  uint32_t next_chunk_v = 0;
  if (lane == 0) next_chunk_v = atomicAdd(item_counter, G);
  uint32_t pf_item = item, pf_chunk_end = item + G;
#define HS_ADVANCE_PF()                                                                  \
  {                                                                                      \
    ++pf_item;                                                                           \
    if (pf_item == pf_chunk_end) {                                                       \
      pf_item = first_dynamic + __builtin_amdgcn_readfirstlane(next_chunk_v);            \
      pf_chunk_end = pf_item + G;                                                        \
      if (lane == 0 && pf_item < n_items) next_chunk_v = atomicAdd(item_counter, G);     \
    }                                                                                    \
  }
  // Descriptors of the current item, the next one and the one after (whose members are fetched now) in
  // scalar registers.  They arrive through VECTOR loads (every lane the same address), issued one item
  // before they are read: scalar loads share their counter with the LDS lookups of the operand build, and
  // the two complete out of order, so every wait for a lookup would also wait for a descriptor still on its
  // way from memory (the descriptor array is streamed once: it never hits the scalar cache); vector loads
  // count in issue order with the member loads, and a descriptor requested a whole item ago has arrived
  // when the members of the item before it have.
  uint32_t vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));  // a zero the compiler cannot see through: keeps the loads vector loads
  const uint4* const desc_v = desc + vzero;
  uint4 d0 = uniform4(desc[2 * (uint64_t)item]), d1 = uniform4(desc[2 * (uint64_t)item + 1]);
  HS_ADVANCE_PF()
  uint32_t next_item = pf_item;
  uint4 nd0 = d0, nd1 = d1;
  if (next_item < n_items) {
    nd0 = uniform4(desc[2 * (uint64_t)next_item]);
    nd1 = uniform4(desc[2 * (uint64_t)next_item + 1]);
  }
  HS_ADVANCE_PF()
  uint32_t nn_item = pf_item;  // the item after next: its descriptor is on its way in dv0 / dv1
  uint4 dv0, dv1;
  {
    const uint64_t ix_ = nn_item < n_items ? nn_item : item;
    dv0 = desc_v[2 * ix_];
    dv1 = desc_v[2 * ix_ + 1];
  }
  // Member loads: lane (n, q) of row tile t takes entry e0 + 16 t + n of the packed array (q = 0, 1: the
  // two quarters need the same 16 bytes), of the record array (q = 2), or -- q = 3 -- the constant factors
  // of the gamma slots, from a block that repeats them 128 times so that the same offsets apply.  One
  // 64-bit base per lane and item, eight loads at immediate offsets.  No clamping at a bucket's ragged
  // end: the rows past it read the next bucket's entries (the arrays are padded by 128 entries) and
  // are masked when survivors are emitted.
  // row 0, 1, 3: j = 0, 1, 2 -- positions 8 j .. 8 j + 7 start at bit 40 j of the packed word: dword j, bit 8 j
  // (eight bytes from dword j on; loading from BYTE 5 j instead, so that no lane shifts anything into place,
  // works and measured 1 % slower: unaligned loads); row 2: the member's four-byte record (and the next one's)
  const bool row2 = row == 2;
  const uint32_t jj = row == 3 ? 2u : (uint32_t)row;
  const char* const lane_base = row2 ? reinterpret_cast<const char*>(rho_base) + 4 * n
                                     : reinterpret_cast<const char*>(packed_base) + 4u * jj + 16 * n;
  const uint32_t lane_shift = row2 ? 2u : 4u;     // bytes per entry of the lane's array: 4 | 16
  const uint32_t lane_step = row2 ? 64u : 256u;   // ... per row tile of 16 members
  J8rLane lc;
  lc.row2 = row2;
  lc.sh = row2 ? 0u : 8u * jj;
  lc.o0 = 0u;
  lc.o1 = row2 ? 0u : 10u;
  lc.w01 = row2 ? 11u : 10u;
  lc.sl = row2 ? 3u : 6u;
  const uint32_t rc = 8u * ((uint32_t)lane & 7u);
  lc.b0 = row2 ? lay.dig_at : rc;
  lc.b1 = row2 ? lay.dig2_at : rc;
  lc.w3 = row2 ? 0u : 10u;
  lc.b2 = row2 ? lay.tab1_at : rc;
  lc.b3 = row2 ? lay.const_at + 8u : rc;
  lc.mrem = row2 ? 0x7fu : 0u;
  const char* const sPairB = reinterpret_cast<const char*>(sLds);
#define HS_LOAD_MEMBERS_R(MK, D0)                                                                   \
  {                                                                                                 \
    const int64_t e0_ = (int64_t)(((uint64_t)(D0).y << 32) | (uint64_t)(D0).x) + (int64_t)(D0).w * 128; \
    const char* p_ = lane_base + (e0_ << lane_shift);                                               \
    _Pragma("unroll") for (int t = 0; t < 8; ++t) MK[t] = *reinterpret_cast<const uint2*>(p_ + lane_step * t); \
  }
  uint2 mkA[8], mkB[8];
  intx4 Bq[2][2][2];
  HS_LOAD_MEMBERS_R(mkA, d0)
  HS_LOAD_MEMBERS_R(mkB, nd0)
  uint32_t cur_qoff = 0xffffffffu;
  bool has_next;
#define HS_ITEM_STEP(MK, MKN)                                                                                  \
  {                                                                                                            \
    const uint32_t M = d0.z, mt = d0.w;                                                                        \
    const uint32_t qoff = d1.x, nQ = d1.z, mstart = d1.w;                                                      \
    const uint32_t wbase = mt * 128u;                                                                          \
    has_next = next_item < n_items;                                                                            \
    /* the descriptor requested one item ago (= this item's once more when there is no item after next) */    \
    const uint4 nnd0 = uniform4(dv0), nnd1 = uniform4(dv1);                                                    \
    HS_ADVANCE_PF() /* pf_item = three items on: its descriptor is requested now */                            \
    {                                                                                                          \
      const uint64_t ix_ = pf_item < n_items ? pf_item : (next_item < n_items ? next_item : item);             \
      dv0 = desc_v[2 * ix_];                                                                                   \
      dv1 = desc_v[2 * ix_ + 1];                                                                               \
    }                                                                                                          \
    if (qoff != cur_qoff) { /* a new segment: its query rows into registers, and wait for them HERE */         \
      cur_qoff = qoff;                                                                                         \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                          \
        const uint32_t qu_ = 32u * u < nQ ? 32u * u : 0u;                                                      \
        load_btile_r(Bq[u], c8t, qoff + qu_, min(32u, nQ - qu_), lane);                                        \
      }                                                                                                        \
      _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                            \
        asm volatile("" ::"v"(Bq[u][0][0]), "v"(Bq[u][0][1]), "v"(Bq[u][1][0]), "v"(Bq[u][1][1]));             \
    }                                                                                                          \
    /* members of the item after next: their address now, the loads when MK has been consumed */             \
    const int64_t e2_ = (int64_t)(((uint64_t)nnd0.y << 32) | (uint64_t)nnd0.x) + (int64_t)nnd0.w * 128;         \
    const char* nm_ = lane_base + (e2_ << lane_shift);                                                         \
    const uint32_t nct_ = (nQ + 15u) >> 4;                                                                     \
    uint32_t gm_;                                                                                              \
    /* AP holds the lookups of this item's first half, issued one half-step ago: complete it (the wait for     \
       them sits here, with nothing younger in flight: the LDS counter has four bits, sixteen younger lookups  \
       could not be told apart from them), send the second half's lookups out, compute the first half */       \
    join8r_finish(AP, MK, 0, lc);                                                                          \
    join8r_lookup(AQ, MK, 1, sPairB, lc);                                                                  \
    switch (nct_) {                                                                                            \
      case 1: gm_ = join8r_half<1, 0>(AP, Bq, smask); break;                                                   \
      case 2: gm_ = join8r_half<2, 0>(AP, Bq, smask); break;                                                   \
      case 3: gm_ = join8r_half<3, 0>(AP, Bq, smask); break;                                                   \
      default: gm_ = join8r_half<4, 0>(AP, Bq, smask); break;                                                  \
    }                                                                                                          \
    /* MK's first half is consumed: the first half of the members of the item after next */                    \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) MK[t] = *reinterpret_cast<const uint2*>(nm_ + lane_step * t); \
    /* second half: complete, send out the NEXT item's first half (its members were requested one item ago;    \
       the same item's when there is none), compute */                                                         \
    join8r_finish(AQ, MK, 1, lc);                                                                          \
    join8r_lookup(AP, MKN, 0, sPairB, lc);                                                                 \
    switch (nct_) {                                                                                            \
      case 1: gm_ |= join8r_half<1, 1>(AQ, Bq, smask); break;                                                  \
      case 2: gm_ |= join8r_half<2, 1>(AQ, Bq, smask); break;                                                  \
      case 3: gm_ |= join8r_half<3, 1>(AQ, Bq, smask); break;                                                  \
      default: gm_ |= join8r_half<4, 1>(AQ, Bq, smask); break;                                                 \
    }                                                                                                          \
    /* ... and the second half */                                                                              \
    _Pragma("unroll") for (int t = 4; t < 8; ++t) MK[t] = *reinterpret_cast<const uint2*>(nm_ + lane_step * t); \
    if (gm_)                                                                                                   \
      join8r_emit(gm_, (int)((nct_ + 1u) >> 1), smask, qoff, nQ, wbase, M, mstart, lane, res_base, res_used,   \
                  prov_count, prov_cap, prov);                                                                 \
    item = next_item;                                                                                          \
    next_item = nn_item;                                                                                       \
    nn_item = pf_item;                                                                                         \
    d0 = nd0;                                                                                                  \
    d1 = nd1;                                                                                                  \
    nd0 = nnd0;                                                                                                \
    nd1 = nnd1;                                                                                                \
  }
  intx4 AP[4][2], AQ[4][2];
  join8r_lookup(AP, mkA, 0, sPairB, lc);  // the first item's first half
  for (;;) {
    HS_ITEM_STEP(mkA, mkB)
    if (!has_next) break;
    HS_ITEM_STEP(mkB, mkA)
    if (!has_next) break;
  }
#undef HS_ITEM_STEP
#undef HS_LOAD_MEMBERS_R
#undef HS_ADVANCE_PF
  close_reservation(prov, res_base, res_used, prov_cap, lane);
}

// Survivors of the 4-column int8 bound, refined with all 8 columns before the exact fp64 decision:
// d2 = |x|^2 + |c|^2 - 2 (x_A.c_A + x_B.c_B) and, per half, s^2 x.c <= x^.c^ + L1(x^)/2 + L1(c^)/2 +
// dims/4 + saturation penalty (the join's own inequality), so a pair with d2 <= R^2 satisfies
//   |x|^2 + (|c|^2 - R^2) <= 2 [ (x^A.c^A + E_A) / s_A^2 + (x^B.c^B + E_B) / s_B^2 ].
// One survivor per lane: 25 table rows (LDS), 2 x 100 query bytes, two v_dot4 per position.  The
// 4-column bound passes ~10 pairs per true hit, this one ~1.1: hs_finalize_kernel, whose cost is
// per survivor, gets a list 7-8 times shorter.  One-sided like every filter here.
template <int PW>
__global__ __launch_bounds__(256) void hs_refine8_kernel(hs_tables_dev tabs,
                                                         const uint2* __restrict__ prov,
                                                         const uint32_t* __restrict__ prov_count,
                                                         uint32_t prov_cap,
                                                         const uint32_t* __restrict__ sorted_ql,
                                                         const int8_t* __restrict__ c8,
                                                         const int8_t* __restrict__ c8b,
                                                         const uint4* __restrict__ tabR,
                                                         const float* __restrict__ scale, int k, int L,
                                                         const uint32_t* __restrict__ qstart,
                                                         const uint32_t* __restrict__ qcount,
                                                         uint2* __restrict__ out,
                                                         uint32_t* __restrict__ out_count) {
  constexpr int NPOS = 25 * PW;              // positions a packed k-mer can hold
  constexpr int NPC = (NPOS + 3) / 4;        // 16-byte pieces of a row that carry coordinates
  __shared__ uint4 sTabR[32];
  __shared__ uint2 s_keep[256];
  __shared__ uint32_t s_n, s_base;
  const int tid = threadIdx.x;
  if (tid < 32) sTabR[tid] = tabR[tid];
  if (tid == 0) s_n = 0;
  __syncthreads();
  const uint32_t n = min(*prov_count, prov_cap);
  const int ROW = 32 * ks_of(k);
  const double sA = (double)scale[0], sB = (double)scale[2];
  const double iA = 1.0 / (sA * sA), iB = 1.0 / (sB * sB);
  const double dims4 = 0.25 * (double)(QD * k);
  for (uint32_t base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u) {
    const uint32_t e = base + (uint32_t)tid;
    uint32_t ql = e < n ? prov[e].x : 0xffffffffu;
    const uint32_t pos = e < n ? prov[e].y : 0u;
    const bool live = ql != 0xffffffffu;  // unused slot of a wave's reserved block
    if (live && (ql & HS_PROV_INDIRECT)) ql = sorted_ql[ql & ~HS_PROV_INDIRECT];
    bool pass = false;
    uint32_t fs_q = 0, fs_id = 0;
    int fs_l = 0;
    if (live) {
      const uint32_t q = ql / (uint32_t)L, l = ql % (uint32_t)L;
      const uint4 pk = tabs.t[l].packed[(uint64_t)pos * PW];
      uint4 pk1 = pk;
      if constexpr (PW == 2) pk1 = tabs.t[l].packed[(uint64_t)pos * PW + 1];
      const Stream256 st = stitch<PW>(pk, pk1);
      const char* rowa = reinterpret_cast<const char*>(c8) + (uint64_t)q * ROW;
      const char* rowb = reinterpret_cast<const char*>(c8b) + (uint64_t)q * ROW;
      const uint4* ra = reinterpret_cast<const uint4*>(rowa);
      const uint4* rb = reinterpret_cast<const uint4*>(rowb);
      int A[4 * NPC], B[4 * NPC];
#pragma unroll
      for (int g = 0; g < NPC; ++g) {
        const uint4 va = ra[g], vb = rb[g];
        A[4 * g] = (int)va.x; A[4 * g + 1] = (int)va.y; A[4 * g + 2] = (int)va.z; A[4 * g + 3] = (int)va.w;
        B[4 * g] = (int)vb.x; B[4 * g + 1] = (int)vb.y; B[4 * g + 2] = (int)vb.z; B[4 * g + 3] = (int)vb.w;
      }
      const double cq = *reinterpret_cast<const double*>(rowb + ROW - 24);
      const float eA = *reinterpret_cast<const float*>(rowb + ROW - 16);
      const float eB = *reinterpret_cast<const float*>(rowb + ROW - 12);
      int dotA = 0, dotB = 0;
      uint32_t l1 = 0;
      float nx = 0.f;
#define HS_R(P)                                                                      \
  if ((P) < NPOS && (P) < k) {                                                       \
    const uint4 row = sTabR[stream_at<5 * (P)>(st)];                                 \
    dotA = __builtin_amdgcn_sdot4((int)row.x, A[(P) < NPOS ? (P) : 0], dotA, false); \
    dotB = __builtin_amdgcn_sdot4((int)row.y, B[(P) < NPOS ? (P) : 0], dotB, false); \
    nx += __uint_as_float(row.z);                                                    \
    l1 += row.w;                                                                     \
  }
#define HS_R10(P) HS_R(P) HS_R(P + 1) HS_R(P + 2) HS_R(P + 3) HS_R(P + 4) HS_R(P + 5) HS_R(P + 6) HS_R(P + 7) HS_R(P + 8) HS_R(P + 9)
      HS_R10(0) HS_R10(10) HS_R(20) HS_R(21) HS_R(22) HS_R(23) HS_R(24)
      if constexpr (PW == 2) { HS_R10(25) HS_R10(35) HS_R(45) HS_R(46) HS_R(47) HS_R(48) HS_R(49) }
#undef HS_R10
#undef HS_R
      const double uA = ((double)dotA + 0.5 * (double)(l1 & 0xffffu) + (double)eA + dims4) * iA;
      const double uB = ((double)dotB + 0.5 * (double)(l1 >> 16) + (double)eB + dims4) * iB;
      const double lhs = (double)nx + cq, rhs = 2.0 * (uA + uB);
      // margin: fp32 sums of the row norms (k x 6e-8 relative) and the roundings of this line
      pass = !(lhs > rhs + 1e-5 * ((double)nx + fabs(cq) + 1.0));
      fs_q = q;
      fs_l = (int)l;
      if (pass && l) fs_id = tabs.t[l].ids[pos];
    }
    // first-seen rule (label[], motif_both_points.cpp:233), which does not depend on the distance: a pair
    // whose k-mer sits in the probed bucket of an EARLIER table is never reported here (the whole wave calls:
    // the tables are looked at a few at a time, for the lanes that still need them)
    if (__ballot(pass && fs_l > 0)) pass = pass && !seen_in_earlier_table(tabs, qstart, qcount, fs_q, fs_l, L, fs_id, pass);
    // block-level compaction: one access to the global counter per block and round (same-address
    // atomics deliver ~90 per microsecond: one per wave and round took longer than the arithmetic)
    if (pass) s_keep[atomicAdd(&s_n, 1u)] = make_uint2(ql, pos);
    __syncthreads();
    const uint32_t kept = s_n;
    if (tid == 0 && kept) s_base = atomicAdd(out_count, kept);
    __syncthreads();
    if ((uint32_t)tid < kept) out[s_base + (uint32_t)tid] = s_keep[tid];
    __syncthreads();
    if (tid == 0) s_n = 0;
    __syncthreads();
  }
}

inline unsigned blocks_for(uint64_t n, unsigned per = 256) { return (unsigned)((n + per - 1) / per); }

}  // namespace

hipError_t hs_launch_jtables8(const double* d_coords, int alphabet, void* d_tab8, float* d_scale,
                              uint32_t* d_unsafe, void* d_tabR, void* d_tabW, hipStream_t s) {
  hs_jtables8_kernel<<<1, 32, 0, s>>>(d_coords, alphabet, (uint4*)d_tab8, d_scale, d_unsafe, (uint4*)d_tabR,
                                      (uint4*)d_tabW);
  return hipGetLastError();
}

hipError_t hs_launch_qprep8(const double* d_centers, uint32_t nq, int k, int wide, double r2,
                            const float* d_scale, void* d_c8, uint32_t* d_unsafe, void* d_c8b,
                            hipStream_t s) {
  if (!nq) return hipSuccess;
  if (wide)
    hs_qprep8_kernel<true><<<blocks_for(nq, 4), 256, 0, s>>>(d_centers, nq, k, r2, d_scale, (int8_t*)d_c8,
                                                             d_unsafe, nullptr);
  else
    hs_qprep8_kernel<false><<<blocks_for(nq, 4), 256, 0, s>>>(d_centers, nq, k, r2, d_scale, (int8_t*)d_c8,
                                                              d_unsafe, (int8_t*)d_c8b);
  return hipGetLastError();
}

hipError_t hs_launch_qprep8_codes(const uint8_t* d_qcodes, uint32_t nq, int k, int wide, double r2,
                                  const double* d_coords, const void* d_tab8, const void* d_tabR,
                                  const void* d_tabW, const float* d_scale, void* d_c8, void* d_c8b,
                                  hipStream_t s) {
  if (!nq) return hipSuccess;
  if (wide)
    hs_qprep8_codes_kernel<true><<<blocks_for(nq), 256, 0, s>>>(d_qcodes, nq, k, r2, d_coords, (const uint4*)d_tab8,
                                                                (const uint4*)d_tabR, (const uint4*)d_tabW,
                                                                d_scale, (int8_t*)d_c8, nullptr);
  else
    hs_qprep8_codes_kernel<false><<<blocks_for(nq), 256, 0, s>>>(d_qcodes, nq, k, r2, d_coords, (const uint4*)d_tab8,
                                                                 (const uint4*)d_tabR, (const uint4*)d_tabW,
                                                                 d_scale, (int8_t*)d_c8, (int8_t*)d_c8b);
  return hipGetLastError();
}

int hs_join8_row_bytes(int k, int wide) { return 32 * ks_of(k, wide != 0); }
// 128 members where a 16x16x64 kernel runs the items (hs_join8x_kernel, hs_join8r_kernel, hs_join8xw_kernel),
// 64 for the 32x32x32 forms of the longer rows
uint32_t hs_join8_members_per_item(int k, int wide) {
  const int KS = ks_of(k, wide != 0);
  return (KS == 4 || (wide && KS == 6)) ? 128u : 64u;
}

hipError_t hs_launch_gather_c8t(const void* d_c8, const uint32_t* d_sorted_ql, const uint32_t* d_seg_qoff,
                                const uint32_t* d_seg_of, uint32_t nql, int L, int k, int wide, void* d_out,
                                hipStream_t s) {
  if (!nql) return hipSuccess;
  hs_gather_c8t_kernel<<<blocks_for((uint64_t)nql * 8), 256, 0, s>>>((const int8_t*)d_c8, d_sorted_ql, d_seg_qoff, d_seg_of, nql,
                                                       L, 2 * ks_of(k, wide != 0), (uint4*)d_out);
  return hipGetLastError();
}

hipError_t hs_launch_join8w(const uint4* d_desc, uint32_t n_items, const uint4* d_packed_base,
                            const uint4* d_rec_base, const void* d_c8t, const void* d_tab8, int k, int wide,
                            uint32_t* d_prov_count, uint32_t prov_cap, uint2* d_prov,
                            uint32_t* d_item_counter, int n_blocks, const uint32_t* d_n_items,
                            double pairs_per_item, uint32_t xcd_run, hipStream_t s, uint32_t chunk) {
  if (!n_items) return hipSuccess;
  // (*d_item_counter is zeroed by the caller, with the batch's other counters)
  // Chunk size = items per access to the global counter.  Same-address atomics complete at only
  // ~ 90 per microsecond on this part, so a chunk has to be worth ~ 100 us of a wave's time: 8 items
  // at C2 (79 k pairs per item), 32-48 where items are small (C3 shape at W = 160: 3.5 k pairs per
  // item, 1.7e7 items -- with chunks of 8 the counter alone took 2.1e6 / 88 = 24 ms of a 26 ms
  // kernel); bigger chunks than that cost balance at the end.  pairs_per_item = the previous
  // batch's average (0: unknown).  Fewer for small launches.
  const uint32_t n_waves = (uint32_t)n_blocks * 4u;
  uint32_t g_max = 8u;
  if (pairs_per_item > 0.0) g_max = (uint32_t)std::max(8.0, std::min(48.0, 7.0e5 / pairs_per_item));
  // up to 8: at least 8 chunks per wave; beyond: at least 64 (the balance at the end is paid in
  // chunks: k = 39 at the C2 sizes, 576 items per wave, lost 3 % with chunks of 18)
  const uint32_t g_small = std::min(8u, n_items / (n_waves * 8u));
  const uint32_t G = chunk ? chunk : std::max(2u, std::min(g_max, std::max(g_small, n_items / (n_waves * 64u))));
  // k <= 25: JT = 4 row tiles per wave (128 members), 4 k-steps, 2 waves per SIMD.  (JT = 2 at 4
  // waves per SIMD, with work items of 64 members, was measured 1.7x slower there: twice the B-tile
  // traffic and per-item work.)  k > 25: 64 members per wave over 6 or 8 k-steps -- the operands of
  // 128 members would not leave registers for the query tiles in flight.
#define HS_J8(JT_, KS_)                                                                               \
  hs_join8w_kernel<JT_, KS_><<<n_blocks, 256, 0, s>>>(d_desc, n_items, d_packed_base, d_rec_base,     \
                                                      (const uint4*)d_c8t, (const uint4*)d_tab8,      \
                                                      d_prov_count, prov_cap, d_prov, d_item_counter, G, \
                                                      d_n_items)
  const int KS = ks_of(k, wide != 0);
  // the 16x16x64 forms where they exist (k <= 25 with 4-column rows, k <= 20 with rows over all 8 columns);
  // the 32x32x32 form for the longer rows
  if (wide && KS == 6)  // d_tab8 = the 8-column table here
    hs_join8xw_kernel<<<n_blocks, 256, 0, s>>>(d_desc, n_items, d_packed_base, d_rec_base, (const uint4*)d_c8t,
                                               (const uint4*)d_tab8, d_prov_count, prov_cap, d_prov, d_item_counter,
                                               G, d_n_items);
  else if (wide)
    hs_join8w_kernel<2, 8, true><<<n_blocks, 256, 0, s>>>(d_desc, n_items, d_packed_base, d_rec_base,
                                                          (const uint4*)d_c8t, (const uint4*)d_tab8, d_prov_count,
                                                          prov_cap, d_prov, d_item_counter, G, d_n_items);
  else if (KS == 4)
    hs_join8x_kernel<<<n_blocks, 256, 0, s>>>(d_desc, n_items, d_packed_base, d_rec_base, (const uint4*)d_c8t,
                                              (const uint4*)d_tab8, d_prov_count, prov_cap, d_prov,
                                              d_item_counter, G, d_n_items, xcd_run);
  else if (KS == 6) HS_J8(2, 6);
  else HS_J8(2, 8);
#undef HS_J8
#ifdef HS_JOIN_TIMING
  {
    unsigned long long t[8];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_join8_timing), sizeof(t));
    fprintf(stderr, "join8 timing (wave-cycles): chunk %llu build %llu next-desc+issue %llu first group %llu other groups %llu flush %llu member-wait %llu | longest wave %llu, mean wave %llu\n",
            t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7],
            (t[0] + t[1] + t[2] + t[3] + t[4] + t[5] + t[6]) / (unsigned long long)(n_blocks * 4));
    memset(t, 0, sizeof(t));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_join8_timing), t, sizeof(t));
  }
#endif
  return hipGetLastError();
}

// The tail [d_split[0], d_split[1]) of the item list -- the segments with at most HS_JR_MAXQ probing
// queries, k <= 25, 4-column rows -- through the query-resident kernel.  d_cn_rep: the constant factors
// of the gamma slots, 128 times over (2 KB).  *d_item_counter zeroed by the caller.
hipError_t hs_launch_join8r(const uint4* d_desc, uint32_t desc_cap, const uint32_t* d_split,
                            const uint4* d_packed_base, const uint32_t* d_rho_base, const void* d_c8t,
                            const void* d_tab8, int alphabet, uint32_t* d_prov_count, uint32_t prov_cap,
                            uint2* d_prov, uint32_t* d_item_counter, int n_blocks, double pairs_per_item,
                            hipStream_t s) {
  if (!desc_cap) return hipSuccess;
  if (alphabet < 1 || alphabet > HS_JR_MAX_ALPHABET) return hipErrorInvalidValue;  // (the caller routes by it)
  const J8rLayout lay = join8r_layout(alphabet);
  {  // more than the 64 KB a kernel may take without asking
    static int granted = 0;  // largest size asked for so far (one value per process is enough: it only grows)
    if ((int)lay.total > granted) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(hs_join8r_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lay.total);
      if (e != hipSuccess) return e;
      granted = (int)lay.total;
    }
  }
  // chunks of consecutive items = consecutive member tiles of one bucket (the query rows stay); sized like
  // hs_launch_join8w's, from the previous batch's pairs per item
  uint32_t G = 16u;
  if (pairs_per_item > 0.0) G = (uint32_t)std::max(8.0, std::min(64.0, 1.4e5 / pairs_per_item * 8.0));
  const uint32_t n_waves = (uint32_t)n_blocks * 4u;
  G = std::max(2u, std::min(G, std::max(2u, desc_cap / (n_waves * 32u))));
  hs_join8r_kernel<<<n_blocks, 256, lay.total, s>>>(d_desc, d_split, desc_cap, d_packed_base, d_rho_base,
                                                    (const uint4*)d_c8t, (const uint4*)d_tab8, alphabet, d_prov_count,
                                                    prov_cap, d_prov, d_item_counter, G);
  return hipGetLastError();
}

hipError_t hs_launch_gather_rec8(const uint4* d_packed_all, const uint32_t* d_ids_sorted, uint32_t n,
                                 int k, int wide, const void* d_tab8, const void* d_tabW, const float* d_scale,
                                 uint4* d_out_packed, uint4* d_out_rec, uint32_t* d_out_rho, hipStream_t s) {
  if (!n) return hipSuccess;
  hs_gather_rec8_kernel<<<blocks_for(n), 256, 0, s>>>(d_packed_all, d_ids_sorted, n, k, hs_packed_words(k), wide,
                                                      (const uint4*)d_tab8, (const uint4*)d_tabW, d_scale,
                                                      d_out_packed, d_out_rec, d_out_rho);
  return hipGetLastError();
}

hipError_t hs_launch_refine8(const hs_tables_dev& tabs, const uint2* d_prov, const uint32_t* d_prov_count,
                             uint32_t prov_cap, const uint32_t* d_sorted_ql, const void* d_c8,
                             const void* d_c8b, const void* d_tabR, const float* d_scale, int k, int L,
                             const uint32_t* d_qstart, const uint32_t* d_qcount,
                             uint2* d_out, uint32_t* d_out_count, hipStream_t s) {
  if (k <= 25)
    hs_refine8_kernel<1><<<1024, 256, 0, s>>>(tabs, d_prov, d_prov_count, prov_cap, d_sorted_ql,
                                              (const int8_t*)d_c8, (const int8_t*)d_c8b, (const uint4*)d_tabR,
                                              d_scale, k, L, d_qstart, d_qcount, d_out, d_out_count);
  else
    hs_refine8_kernel<2><<<1024, 256, 0, s>>>(tabs, d_prov, d_prov_count, prov_cap, d_sorted_ql,
                                              (const int8_t*)d_c8, (const int8_t*)d_c8b, (const uint4*)d_tabR,
                                              d_scale, k, L, d_qstart, d_qcount, d_out, d_out_count);
  return hipGetLastError();
}
