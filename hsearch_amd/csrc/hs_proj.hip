// hs_proj.hip -- the LSH projection on the matrix cores, with the reference's bucket integers kept
// bit for bit (hard part 1; rows a4/a5).
//
// Reference (lsh.hpp:33-49):  dot = sum_i x_i * a_i  strictly left to right in fp64, product and sum
// rounded separately;  t = (dot + b) / W;  bucket = int(floor(t)).  The exact kernel
// (hs_hash_kernel, hs_kernels.hip) evaluates precisely that on the vector ALU: 2 d K L dependent
// fp64 operations per point.  Here the same integers are produced in two steps:
//
//  1. FAST PASS on v_mfma_i32_32x32x32_i8.  Point and plane are quantised to 16-bit fixed point,
//         X_i = rint(x_i 2^ex),  A_i = rint(a_i 2^ea),   |X_i|, |A_i| <= 32639,
//     split into signed bytes X = 256 X1 + X0, A = 256 A1 + A0, and
//         N = sum_i X_i A_i = 65536 (X1.A1) + 256 (X1.A0 + X0.A1) + (X0.A0)
//     is EXACT integer arithmetic (three int32 accumulators, four MFMA passes per 32-deep k-step).
//     T~ = (N 2^-(ex+ea) + b) / W approximates the reference's t with a PROVEN bound:
//         |x.a - N 2^-(ex+ea)| <= da |x|_1 + dx |a^|_1            (da = 2^-(ea+1), dx = 2^-(ex+1))
//         |dot_ref - x.a|      <= gamma |x|_1 max|a|,  gamma = 1.01 (d+2) 2^-53   (Higham, recursive sum)
//         |t_ref - T|          <= (1 + 2^-50) E1 / W + 2^-51 |T|   (the two roundings of (dot+b)/W)
//     so  |t_ref - T~| <= E := |x|_1 alpha_f + dx beta_f + 2^-49 |T~| + 2^-40  with
//     alpha_f = 1.000001 (da_f + gamma max|a_f|) / W,  beta_f = 1.000001 |a^_f|_1 / W.
//     If [T~ - E, T~ + E] contains no integer, floor(t_ref) = floor(T~): the value is FINAL.
//  2. Otherwise the (point, function) pair is FLAGGED and recomputed by hs_proj_fix_kernel with the
//     reference's own operation sequence (__dmul_rn/__dadd_rn/__ddiv_rn, like hs_hash_kernel).
//
// At W = 200, k = 25 the bound is E ~ 7e-4: ~0.14 % of the values are flagged.  Anything the fixed
// point cannot carry (non-finite or extreme magnitudes, NaN bounds) fails the comparisons and is
// flagged too, so the result never depends on the fast pass being applicable; when the flag list
// overflows, the fix kernel recomputes everything.
//
// Layouts.  MFMA operands (both): lane (r = lane & 31, h = lane >> 5) holds the 16 bytes
// k = 16 h .. 16 h + 15 of row r of a 32-deep k-step; dimension i = 32 s + k, i.e. position
// 4 s + 2 h + (j >> 3), coordinate j & 7.  First operand = planes (rows = functions), second =
// points (columns): output element i of lane (p, h) is function (i & 3) + 8 (i >> 2) + 4 h of point p.
//   aq   [tile][S][2][64] uint4   plane digit bytes in fragment order (u = 0: A1, u = 1: A0)
//   fn   [F]  {2^-ea, b, alpha, beta}
//   tab        codes path: per residue {X1[8], X0[8]} + |row|_1, 2^-ex, dx        (hs_proj_table)
//   xq   [n][S][2 halves][2 digit planes] uint4 + xmeta[n] {|x|_1, 2^-ex, dx}      (points path)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "hs_internal.h"

namespace {

typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx16 __attribute__((ext_vector_type(16)));

constexpr int QMAX = 32639;        // 127 * 256 + 127: both digits of +-QMAX are int8
constexpr uint32_t RES = 128;      // flag slots a wave reserves per counter access

struct ProjFn {
  double sa, b, alpha, beta;
};

// largest e with rint(m * 2^e) <= QMAX  (m = max |value| > 0, finite); *ok = false when out of range
__device__ __forceinline__ int fixed_exponent(double m, bool* ok) {
  int e2;
  (void)frexp(m, &e2);  // m = f * 2^e2, f in [0.5, 1)
  int e = 15 - e2;      // m * 2^e in [2^14, 2^15)
  if (rint(ldexp(m, e)) > (double)QMAX) e -= 1;
  if (e > 200 || e < -200) *ok = false;
  return e;
}

__device__ __forceinline__ void split_digits(int X, int* hi, int* lo) {
  const int l = (int)(int8_t)(X & 0xff);  // signed low byte
  *lo = l;
  *hi = (X - l) >> 8;                     // exact: X - l is a multiple of 256; |hi| <= 127 for |X| <= QMAX
}

// ------------------------------------------------------------------------------------ quantisers
// coordinate table -> tab: one block of 32 threads
__global__ void hs_quant_table_kernel(const double* __restrict__ coords, int alphabet,
                                      hs_proj_table* __restrict__ tab) {
  __shared__ double smax[32];
  __shared__ int sbad[32];
  const int c = threadIdx.x;
  double m = 0.0;
  int bad = 0;
  for (int j = 0; j < 8; ++j) {
    const double v = c < alphabet ? coords[c * 8 + j] : 0.0;
    if (!(fabs(v) < 1e300)) bad = 1;
    m = fmax(m, fabs(v));
  }
  smax[c] = m;
  sbad[c] = bad;
  __syncthreads();
  double mm = 0.0;
  int anybad = 0;
  for (int i = 0; i < 32; ++i) {
    mm = fmax(mm, smax[i]);
    anybad |= sbad[i];
  }
  bool ok = !anybad;
  int ex = 0;
  if (ok && mm > 0.0) ex = fixed_exponent(mm, &ok);
  if (!ok) ex = 0;
  uint32_t w[4] = {0u, 0u, 0u, 0u};
  double l1 = 0.0;
  for (int j = 0; j < 8; ++j) {
    const double v = (ok && c < alphabet) ? coords[c * 8 + j] : 0.0;
    const int X = (int)rint(ldexp(v, ex));
    int hi, lo;
    split_digits(X, &hi, &lo);
    w[j >> 2] |= ((uint32_t)hi & 0xffu) << (8 * (j & 3));
    w[2 + (j >> 2)] |= ((uint32_t)lo & 0xffu) << (8 * (j & 3));
    l1 += fabs(v);
  }
  tab->dig[c] = make_uint4(w[0], w[1], w[2], w[3]);
  tab->l1[c] = l1 * (1.0 + 0x1p-40);
  smax[c] = l1;
  __syncthreads();
  if (c == 0) {
    double lm = 0.0;
    for (int i = 0; i < 32; ++i) lm = fmax(lm, smax[i]);
    tab->sx = ldexp(1.0, -ex);
    tab->dx = ldexp(1.0, -(ex + 1));
    tab->l1max = lm;
    tab->unsafe = ok ? 0u : 1u;
  }
}

// planes a[F][d] (row-major), b[F] -> digit fragments in both tilings + per-function constants.
// One wave per function.  stats[0] = max da, stats[1] = max |a^|_1 (as the bits of positive doubles).
__global__ __launch_bounds__(256) void hs_quant_planes_kernel(const double* __restrict__ a,
                                                              const double* __restrict__ b, int F, int d,
                                                              int K, int S, double W, double eps_scale,
                                                              uint4* __restrict__ aq_all,
                                                              uint4* __restrict__ aq_tab,
                                                              ProjFn* __restrict__ fn,
                                                              unsigned long long* __restrict__ stats) {
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= F) return;
  const int lane = threadIdx.x & 63;
  const double* row = a + (size_t)f * d;
  double m = 0.0;
  bool ok = true;
  for (int i = lane; i < d; i += 64) {
    const double v = row[i];
    if (!(fabs(v) < 1e300)) ok = false;
    m = fmax(m, fabs(v));
  }
  for (int off = 32; off; off >>= 1) m = fmax(m, __shfl_xor(m, off));
  ok = __ballot(!ok) == 0;
  int ea = 0;
  if (ok && m > 0.0) ea = fixed_exponent(m, &ok);
  ok = __ballot(!ok) == 0;
  const double bf = b[f];
  if (!(fabs(bf) < 0x1p200) || !(W > 0x1p-200) || !(W < 0x1p200)) ok = false;
  if (!ok) ea = 0;
  char* all = reinterpret_cast<char*>(aq_all);
  char* tab = reinterpret_cast<char*>(aq_tab);
  const int ft = f >> 5, r = f & 31, tl = f / K, c = f % K;
  long long l1 = 0;
  for (int i = lane; i < 32 * S; i += 64) {
    const double v = (ok && i < d) ? row[i] : 0.0;
    const int A = (int)rint(ldexp(v, ea));
    int hi, lo;
    split_digits(A, &hi, &lo);
    l1 += abs(A);
    const int s = i >> 5, h = (i >> 4) & 1, j = i & 15;
    const size_t o_all = ((((size_t)ft * S + s) * 2) * 64 + (h * 32 + r)) * 16 + j;
    const size_t o_tab = ((((size_t)tl * S + s) * 2) * 64 + (h * 32 + c)) * 16 + j;
    all[o_all] = (char)hi;
    all[o_all + 64 * 16] = (char)lo;
    tab[o_tab] = (char)hi;
    tab[o_tab + 64 * 16] = (char)lo;
  }
  for (int off = 32; off; off >>= 1) l1 += __shfl_xor(l1, off);
  if (lane == 0) {
    const double sa = ldexp(1.0, -ea), da = ldexp(1.0, -(ea + 1));
    const double a1 = (double)l1 * sa;
    const double gamma = 1.01 * (double)(d + 2) * 0x1p-53;
    ProjFn o;
    o.sa = sa;
    o.b = bf;
    // eps_scale (>= 1, tests): inflates the bound, so more values take the exact path -- never fewer
    o.alpha = ok ? eps_scale * 1.000001 * (da + gamma * m) / W : (double)INFINITY;
    o.beta = ok ? eps_scale * 1.000001 * a1 / W : (double)INFINITY;
    fn[f] = o;
    if (ok) {
      atomicMax(&stats[0], (unsigned long long)__double_as_longlong(da));
      atomicMax(&stats[1], (unsigned long long)__double_as_longlong(a1));
    } else {
      atomicMax(&stats[2], 1ull);
    }
  }
}

// points x[n][d] -> digit fragments xq + xmeta {|x|_1 (upper bound), 2^-ex, dx}.  One wave per point.
__global__ __launch_bounds__(256) void hs_quant_points_kernel(const double* __restrict__ pts, uint64_t n,
                                                              int d, int S, uint4* __restrict__ xq,
                                                              double* __restrict__ xmeta) {
  const uint64_t p = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n) return;
  const int lane = threadIdx.x & 63;
  const double* row = pts + p * (uint64_t)d;
  // a lane owns 4 consecutive coordinates (two sets of them for rows of more than 256): the row is
  // read once, 32 bytes per lane, and the digits leave as one dword of high and one of low bytes
  const int nv = 32 * S;
  double v[2][4];
  double m = 0.0, l1 = 0.0;
  bool ok = true;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int base = 4 * (lane + 64 * it);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[it][e] = 0.0;
    if (base + 3 < d) {
      const double2 a = *reinterpret_cast<const double2*>(row + base);
      const double2 b = *reinterpret_cast<const double2*>(row + base + 2);
      v[it][0] = a.x; v[it][1] = a.y; v[it][2] = b.x; v[it][3] = b.y;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (base + e < d) v[it][e] = row[base + e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double a = fabs(v[it][e]);
      if (!(a < 1e300)) ok = false;
      m = fmax(m, a);
      l1 += a;
    }
  }
  for (int off = 32; off; off >>= 1) {
    m = fmax(m, __shfl_xor(m, off));
    l1 += __shfl_xor(l1, off);
  }
  ok = __ballot(!ok) == 0;
  int ex = 0;
  if (ok && m > 0.0) ex = fixed_exponent(m, &ok);
  ok = __ballot(!ok) == 0;
  if (!ok) ex = 0;
  char* out = reinterpret_cast<char*>(xq) + p * (uint64_t)S * 64;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int base = 4 * (lane + 64 * it);
    if (base >= nv) continue;
    uint32_t whi = 0, wlo = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int X = (int)rint(ldexp(ok ? v[it][e] : 0.0, ex));
      int hi, lo;
      split_digits(X, &hi, &lo);
      whi |= ((uint32_t)hi & 0xffu) << (8 * e);
      wlo |= ((uint32_t)lo & 0xffu) << (8 * e);
    }
    const int s = base >> 5, h = (base >> 4) & 1, j = base & 15;
    *reinterpret_cast<uint32_t*>(out + s * 64 + h * 32 + j) = whi;
    *reinterpret_cast<uint32_t*>(out + s * 64 + h * 32 + 16 + j) = wlo;
  }
  if (lane == 0) {
    // the wave sums |x_i| in a tree: <= d roundings, relative error < d 2^-53
    xmeta[3 * p] = ok ? l1 * (1.0 + 0x1p-40) : (double)INFINITY;
    xmeta[3 * p + 1] = ldexp(1.0, -ex);
    xmeta[3 * p + 2] = ldexp(1.0, -(ex + 1));
  }
}

// ---------------------------------------------------------------------------------- fast pass
template <int S, bool FROM_CODES>
__global__ __launch_bounds__(256, S > 10 ? 1 : 2) void hs_proj_kernel(
    const uint8_t* __restrict__ codes, const uint4* __restrict__ xq, const double* __restrict__ xmeta,
    uint64_t n, int k, const uint4* __restrict__ aq, const ProjFn* __restrict__ fn, int F,
    const hs_proj_table* __restrict__ tab, double invW, int32_t* __restrict__ out, int out_stride,
    uint2* __restrict__ flags, uint32_t flag_cap, uint32_t* __restrict__ flag_count) {
  __shared__ uint4 s_dig[32];
  __shared__ double s_l1[32];
  __shared__ ProjFn s_fn[32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int ft = blockIdx.y;           // function tile: functions 32 ft .. 32 ft + 31 (those < F are real)
  if (tid < 32) {
    if (FROM_CODES) {
      s_dig[tid] = tab->dig[tid];
      s_l1[tid] = tab->l1[tid];
    }
    const bool have = 32 * ft + tid < F;  // padding rows: never stored, bounds infinite anyway
    const ProjFn* src = fn + (have ? 32 * ft + tid : 0);
    s_fn[tid].sa = have ? src->sa : 0.0;
    s_fn[tid].b = have ? src->b : 0.0;
    s_fn[tid].alpha = have ? src->alpha : (double)INFINITY;
    s_fn[tid].beta = have ? src->beta : (double)INFINITY;
  }
  __syncthreads();
  // the plane fragments of this function tile stay in registers for all the wave's point tiles
  intx4 Afr[S][2];
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint4 v = aq[(((size_t)ft * S + s) * 2 + u) * 64 + lane];
      Afr[s][u] = intx4{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
    }
  const double sx_tab = FROM_CODES ? tab->sx : 0.0, dx_tab = FROM_CODES ? tab->dx : 0.0;
  const bool tab_unsafe = FROM_CODES && tab->unsafe != 0u;
  const uint64_t n_tiles = (n + 31) / 32;
  // 16-byte stores of the bucket ints need rows that start on 16 bytes
  const bool aligned16 = (out_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
  uint32_t res_base = 0, res_left = 0, flagged = 0;
  // Raw inputs of a point tile: the lane's residues (codes path: positions 4 s + 2 h, 4 s + 2 h + 1)
  // or its digit fragments (points path).  The NEXT tile's are fetched before the current tile's
  // epilogue (codes: before its MFMAs, into a second register set of 2 S dwords), so their latency
  // is not exposed (a wave walks ~6 tiles for a query batch).  The number of loads per step never
  // depends on the tile (a tile past the end re-reads the last).  Long rows (S > 7, points path)
  // would not fit a whole tile's fragments in registers beside the planes: loaded step by step.
  constexpr bool PF_POINTS = !FROM_CODES && S <= 7;
  constexpr bool PF_CODES = FROM_CODES && S <= 10;  // S = 13: the second residue set would spill
  constexpr int NRAW = PF_POINTS ? 2 * S : 1;
  uint32_t c0[S], c1[S], c0n[PF_CODES ? S : 1], c1n[PF_CODES ? S : 1];
  uint4 raw[NRAW];
  const uint64_t tile0 = (uint64_t)blockIdx.x * 4 + wave, tstride = (uint64_t)gridDim.x * 4;
  auto fetch = [&](uint64_t tile) {
    const uint64_t t = tile < n_tiles ? tile : n_tiles - 1;
    const uint64_t p = t * 32 + (uint64_t)r;
    const bool pvalid = p < n;
    if constexpr (FROM_CODES) {
      const uint8_t* row = codes + (pvalid ? p : 0) * (uint64_t)k;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int q0 = 4 * s + 2 * h;
        // positions past the k-mer (and rows past the end) read position 0 and are masked below
        const uint32_t a0 = (uint32_t)row[q0 < k ? q0 : 0], a1 = (uint32_t)row[q0 + 1 < k ? q0 + 1 : 0];
        if constexpr (PF_CODES) {
          c0n[s] = a0;
          c1n[s] = a1;
        } else {
          c0[s] = a0;
          c1[s] = a1;
        }
      }
    } else if constexpr (PF_POINTS) {
      const uint4* xr = xq + (pvalid ? p : 0) * (uint64_t)S * 4 + 2 * h;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        raw[2 * s] = xr[4 * s];
        raw[2 * s + 1] = xr[4 * s + 1];
      }
    }
  };
  if (tile0 < n_tiles && (PF_CODES || PF_POINTS)) fetch(tile0);
  for (uint64_t tile = tile0; tile < n_tiles; tile += tstride) {
    const uint64_t p = tile * 32 + (uint64_t)r;
    const bool pvalid = p < n;
    intx16 acc_hi, acc_mid, acc_lo;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_hi[i] = acc_mid[i] = acc_lo[i] = 0;
    double x1 = 0.0;
    if constexpr (FROM_CODES) {
      const uint8_t* row_now = codes + (pvalid ? p : 0) * (uint64_t)k;
      if constexpr (PF_CODES) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          c0[s] = c0n[s];
          c1[s] = c1n[s];
        }
        fetch(tile + tstride);
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int q0 = 4 * s + 2 * h;
        const bool v0 = pvalid && q0 < k, v1 = pvalid && q0 + 1 < k;
        if constexpr (!PF_CODES) {  // long rows: residues fetched step by step, no register arrays
          c0[s] = (uint32_t)row_now[q0 < k ? q0 : 0];
          c1[s] = (uint32_t)row_now[q0 + 1 < k ? q0 + 1 : 0];
        }
        uint4 d0 = s_dig[c0[s] & 31u], d1 = s_dig[c1[s] & 31u];
        if (!v0) d0 = make_uint4(0u, 0u, 0u, 0u);
        if (!v1) d1 = make_uint4(0u, 0u, 0u, 0u);
        x1 += (v0 ? s_l1[c0[s] & 31u] : 0.0) + (v1 ? s_l1[c1[s] & 31u] : 0.0);
        const intx4 B1 = intx4{(int)d0.x, (int)d0.y, (int)d1.x, (int)d1.y};
        const intx4 B0 = intx4{(int)d0.z, (int)d0.w, (int)d1.z, (int)d1.w};
        acc_hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(Afr[s][0], B1, acc_hi, 0, 0, 0);
        acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Afr[s][0], B0, acc_mid, 0, 0, 0);
        acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Afr[s][1], B1, acc_mid, 0, 0, 0);
        acc_lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(Afr[s][1], B0, acc_lo, 0, 0, 0);
      }
      x1 += __shfl_xor(x1, 32);        // both halves of the point's positions
      x1 *= (1.0 + 0x1p-40);
    } else {
      const uint4* xr = xq + (pvalid ? p : 0) * (uint64_t)S * 4 + 2 * h;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        uint4 v1, v0;
        if constexpr (PF_POINTS) {
          v1 = raw[2 * s];
          v0 = raw[2 * s + 1];
        } else {
          v1 = xr[4 * s];
          v0 = xr[4 * s + 1];
        }
        const intx4 B1 = intx4{(int)v1.x, (int)v1.y, (int)v1.z, (int)v1.w};
        const intx4 B0 = intx4{(int)v0.x, (int)v0.y, (int)v0.z, (int)v0.w};
        acc_hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(Afr[s][0], B1, acc_hi, 0, 0, 0);
        acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Afr[s][0], B0, acc_mid, 0, 0, 0);
        acc_mid = __builtin_amdgcn_mfma_i32_32x32x32_i8(Afr[s][1], B1, acc_mid, 0, 0, 0);
        acc_lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(Afr[s][1], B0, acc_lo, 0, 0, 0);
      }
      fetch(tile + tstride);  // the registers just consumed take the next tile's fragments
    }
    double sx, dx;
    if (FROM_CODES) {
      sx = sx_tab;
      dx = dx_tab;
      if (tab_unsafe) x1 = (double)INFINITY;
    } else {
      const double* mp = xmeta + 3 * (pvalid ? p : 0);
      x1 = mp[0];
      sx = mp[1];
      dx = mp[2];
    }
    // ---- epilogue: T~, its bound, the certain floor or a flag
    uint32_t fmask = 0;
    int32_t fv[16];
    int32_t* orow = out + (pvalid ? p : 0) * (uint64_t)out_stride + 32 * ft;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int frow = (i & 3) + 8 * (i >> 2) + 4 * h;
      fv[i] = 0;
      if (32 * ft + 8 * (i >> 2) >= F) continue;  // wave-uniform: this group of rows is padding
      const ProjFn c = s_fn[frow];
      const double N = fma((double)acc_hi[i], 65536.0, fma((double)acc_mid[i], 256.0, (double)acc_lo[i]));
      const double dot = N * (c.sa * sx);          // exact: powers of two
      const double T = (dot + c.b) * invW;
      const double fl = floor(T);
      const double frac = T - fl;
      const double E = fma(x1, c.alpha, fma(dx, c.beta, fma(fabs(T), 0x1p-49, 0x1p-40)));
      const bool certain = (frac >= E) && ((1.0 - frac) > E);  // false for NaN / infinite bounds
      const bool real = pvalid && 32 * ft + frow < F;
      fv[i] = (int32_t)fl;
      if (real && !certain) fmask |= 1u << i;
    }
    // the lane's values are 4 groups of 4 consecutive functions (8 j + 4 h + {0..3}): one 16-byte store
    // per group where the whole group is real (F is a multiple of 4 for every K the reference uses)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f0 = 32 * ft + 8 * j + 4 * h;
      if (!pvalid || f0 >= F) continue;
      int32_t* dst = orow + 8 * j + 4 * h;
      if (f0 + 3 < F && aligned16) {
        *reinterpret_cast<int4*>(dst) = make_int4(fv[4 * j], fv[4 * j + 1], fv[4 * j + 2], fv[4 * j + 3]);
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m)
          if (f0 + m < F) dst[m] = fv[4 * j + m];
      }
    }
    if (__ballot(fmask != 0)) {
      // entries of the tile: (point, function); slots from the wave's reservation
      uint32_t total = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) total += (uint32_t)__popcll(__ballot((fmask >> i) & 1u));
      if (total > res_left) {
        // give the rest of the old reservation back as empty slots, take a new one
        for (uint32_t t = (uint32_t)lane; t < res_left; t += 64)
          if (res_base + t < flag_cap) flags[res_base + t] = make_uint2(0xffffffffu, 0u);
        const uint32_t want = total > RES ? total : RES;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(flag_count, want);
        res_base = __builtin_amdgcn_readfirstlane(base);
        res_left = want;
      }
      uint32_t used = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool mine = (fmask >> i) & 1u;
        const unsigned long long bal = __ballot(mine);
        if (mine) {
          const uint32_t o = res_base + used + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
          if (o < flag_cap) flags[o] = make_uint2((uint32_t)p, (uint32_t)(32 * ft + (i & 3) + 8 * (i >> 2) + 4 * h));
        }
        used += (uint32_t)__popcll(bal);
      }
      res_base += total;
      res_left -= total;
      flagged += total;
    }
  }
  if (lane == 0 && flagged) atomicAdd(flag_count + 1, flagged);  // statistics: real entries
  for (uint32_t t = (uint32_t)lane; t < res_left; t += 64)
    if (res_base + t < flag_cap) flags[res_base + t] = make_uint2(0xffffffffu, 0u);
}

// ------------------------------------------------------------------------------------ exact pass
// One lane per flagged (point, function): the reference's operation sequence.  When the list
// overflowed (count > cap) every (point, function) pair is recomputed instead.
template <bool FROM_CODES>
__global__ __launch_bounds__(256) void hs_proj_fix_kernel(const uint8_t* __restrict__ codes,
                                                          const double* __restrict__ pts, uint64_t n, int k,
                                                          const double* __restrict__ aT, int ldf,
                                                          const double* __restrict__ b, int F, double W,
                                                          const double* __restrict__ coords,
                                                          int32_t* __restrict__ out, int out_stride,
                                                          const uint2* __restrict__ flags, uint32_t flag_cap,
                                                          const uint32_t* __restrict__ flag_count) {
  __shared__ double s_coords[HS_ALPHABET_PAD * 8];
  if (FROM_CODES) {
    for (int t = threadIdx.x; t < HS_ALPHABET_PAD * 8; t += 256) s_coords[t] = coords[t];
    __syncthreads();
  }
  const uint32_t count = *flag_count;
  const bool all = count > flag_cap;
  const uint64_t total = all ? n * (uint64_t)F : (uint64_t)count;
  const int d = 8 * k;
  for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (uint64_t)gridDim.x * 256) {
    uint64_t p;
    uint32_t f;
    if (all) {
      p = e / (uint64_t)F;
      f = (uint32_t)(e % (uint64_t)F);
    } else {
      const uint2 en = flags[e];
      if (en.x == 0xffffffffu) continue;
      p = en.x;
      f = en.y;
    }
    const double* col = aT + f;
    double acc = 0.0;
    // 8 dimensions (one position) at a time: the 16 loads are independent and issue together, only
    // the additions form the chain the reference prescribes
    if (FROM_CODES) {
      const uint8_t* row = codes + p * (uint64_t)k;
      for (int pos = 0; pos < k; ++pos) {
        const int c = row[pos] & (HS_ALPHABET_PAD - 1);
        double av[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) av[j] = col[(size_t)(8 * pos + j) * ldf];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __dadd_rn(acc, __dmul_rn(s_coords[c * 8 + j], av[j]));
      }
    } else {
      const double* row = pts + p * (uint64_t)d;
      for (int i0 = 0; i0 < d; i0 += 8) {
        double xv[8], av[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xv[j] = row[i0 + j];
          av[j] = col[(size_t)(i0 + j) * ldf];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __dadd_rn(acc, __dmul_rn(xv[j], av[j]));
      }
    }
    out[p * (uint64_t)out_stride + f] = (int32_t)floor(__ddiv_rn(__dadd_rn(acc, b[f]), W));
  }
}

inline unsigned blocks_for(uint64_t n, unsigned per = 256) { return (unsigned)((n + per - 1) / per); }

}  // namespace

int hs_proj_steps(int k) {
  const int need = (8 * k + 31) / 32;
  return need <= 4 ? 4 : need <= 7 ? 7 : need <= 10 ? 10 : need <= 13 ? 13 : 0;  // 0: not supported
}

hipError_t hs_launch_quant_table(const double* d_coords, int alphabet, hs_proj_table* d_tab, hipStream_t s) {
  hs_quant_table_kernel<<<1, 32, 0, s>>>(d_coords, alphabet, d_tab);
  return hipGetLastError();
}

hipError_t hs_launch_quant_planes(const double* d_a, const double* d_b, int F, int d, int K, int S, double W,
                                  double eps_scale, void* d_aq_all, void* d_aq_tab, void* d_fn,
                                  unsigned long long* d_stats, hipStream_t s) {
  hs_quant_planes_kernel<<<blocks_for((uint64_t)F, 4), 256, 0, s>>>(d_a, d_b, F, d, K, S, W, eps_scale,
                                                                    (uint4*)d_aq_all, (uint4*)d_aq_tab,
                                                                    (ProjFn*)d_fn, d_stats);
  return hipGetLastError();
}

hipError_t hs_launch_quant_points(const double* d_pts, uint64_t n, int k, int S, void* d_xq, double* d_xmeta,
                                  hipStream_t s) {
  if (!n) return hipSuccess;
  hs_quant_points_kernel<<<blocks_for(n, 4), 256, 0, s>>>(d_pts, n, 8 * k, S, (uint4*)d_xq, d_xmeta);
  return hipGetLastError();
}

hipError_t hs_launch_proj(const uint8_t* d_codes, const void* d_xq, const double* d_xmeta, uint64_t n, int k,
                          int S, const void* d_aq, const void* d_fn, int F, const hs_proj_table* d_tab,
                          double W, int32_t* d_out, int out_stride, uint2* d_flags, uint32_t flag_cap,
                          uint32_t* d_flag_count, int n_cu, hipStream_t s) {
  if (!n || !F) return hipSuccess;
  const unsigned ftiles = (unsigned)((F + 31) / 32);
  const uint64_t n_tiles = (n + 31) / 32;
  // persistent in x: every wave keeps its function tile's planes in registers over its point tiles
  const unsigned bx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_tiles + 3) / 4,
                                                                      std::max(1u, (unsigned)(n_cu * 2) / ftiles)));
  const dim3 grid(bx, ftiles);
  const double invW = 1.0 / W;
  const bool from_codes = d_codes != nullptr;
#define HS_PROJ(SS)                                                                                          \
  if (from_codes)                                                                                            \
    hs_proj_kernel<SS, true><<<grid, 256, 0, s>>>(d_codes, nullptr, nullptr, n, k, (const uint4*)d_aq,       \
                                                  (const ProjFn*)d_fn, F, d_tab, invW, d_out, out_stride,    \
                                                  d_flags, flag_cap, d_flag_count);                          \
  else                                                                                                       \
    hs_proj_kernel<SS, false><<<grid, 256, 0, s>>>(nullptr, (const uint4*)d_xq, d_xmeta, n, k,              \
                                                   (const uint4*)d_aq, (const ProjFn*)d_fn, F, d_tab, invW,  \
                                                   d_out, out_stride, d_flags, flag_cap, d_flag_count);
  switch (S) {
    case 4: HS_PROJ(4) break;
    case 7: HS_PROJ(7) break;
    case 10: HS_PROJ(10) break;
    case 13: HS_PROJ(13) break;
    default: return hipErrorInvalidValue;
  }
#undef HS_PROJ
  return hipGetLastError();
}

hipError_t hs_launch_proj_fix(const uint8_t* d_codes, const double* d_pts, uint64_t n, int k, const double* d_aT,
                              int ldf, const double* d_b, int F, double W, const double* d_coords,
                              int32_t* d_out, int out_stride, const uint2* d_flags, uint32_t flag_cap,
                              const uint32_t* d_flag_count, hipStream_t s) {
  if (!n || !F) return hipSuccess;
  if (d_codes)
    hs_proj_fix_kernel<true><<<1024, 256, 0, s>>>(d_codes, nullptr, n, k, d_aT, ldf, d_b, F, W, d_coords, d_out,
                                                  out_stride, d_flags, flag_cap, d_flag_count);
  else
    hs_proj_fix_kernel<false><<<1024, 256, 0, s>>>(nullptr, d_pts, n, k, d_aT, ldf, d_b, F, W, d_coords, d_out,
                                                   out_stride, d_flags, flag_cap, d_flag_count);
  return hipGetLastError();
}
