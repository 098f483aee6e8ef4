// Microbenchmark: int8 MFMA throughput (and the clock the chip holds) on RANDOM operands for the
// two shapes, 2 waves per SIMD, 64 accumulator registers per wave, with a sign-test epilogue.
//   shape 0: v_mfma_i32_32x32x32_i8, 4 row tiles x 1 column tile, K = 128 (16 MFMAs per tile)
//   shape 1: v_mfma_i32_16x16x64_i8, 8 row tiles x 2 column tiles, K = 128 (32 MFMAs per tile)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// SLEEP > 0: each wave idles 64 * SLEEP cycles per tile, so the matrix pipe is busy only part of the
// time (the join kernel's regime: ~65 %): does the shape still matter when the loop is not saturated?
template <int SHAPE, int SLEEP>
__global__ __launch_bounds__(256, 2) void k(int iters, const uint4* __restrict__ rnd, int* out,
                                            unsigned long long* cyc) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  intx4 A[16], B[4];
  for (int i = 0; i < 16; ++i) {
    const uint4 v = rnd[(gid * 16 + i) & 0xfffff];
    A[i] = intx4{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
  }
  for (int i = 0; i < 4; ++i) {
    const uint4 v = rnd[(gid * 4 + i + 77777) & 0xfffff];
    B[i] = intx4{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
  }
  uint32_t keep = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    uint32_t sall = 0xffffffffu;
    if (SHAPE == 0) {
      intx16 acc[4];
      for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[t * 4 + s], B[s], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) sall &= (uint32_t)acc[t][i];
    } else {
      intx4 acc[16];  // 8 row tiles x 2 column tiles of 16x16
      for (int t = 0; t < 16; ++t) acc[t] = intx4{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 16; ++t)
          acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[(t >> 1) * 2 + s], B[(t & 1) * 2 + s], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) sall &= (uint32_t)acc[t][i];
    }
    if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
    if (__ballot((int)sall >= 0) == 0x123456789ull) keep += sall;
    B[0][0] ^= (int)(sall & 0x01010101u);  // loop-carried, keeps the data moving
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (keep == 0xdeadbeef) out[threadIdx.x] = (int)keep;
  if ((threadIdx.x & 63) == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));
}

int main() {
  int* out; unsigned long long* cyc; uint4* rnd;
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&rnd, (1 << 20) * 16));
  uint32_t* h = (uint32_t*)malloc((1 << 20) * 16);
  uint64_t x = 88172645463325252ull;
  for (int i = 0; i < (1 << 22); ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)(x >> 16); }
  CK(hipMemcpy(rnd, h, (1 << 20) * 16, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 40000;
  for (int rep = 0; rep < 3; ++rep)
    for (int var = 0; var < 6; ++var) {
      const int shape = var & 1, sl = var >> 1;
      CK(hipMemset(cyc, 0, 8));
      CK(hipEventRecord(e0));
      switch (var) {
        case 0: k<0, 0><<<512, 256>>>(iters, rnd, out, cyc); break;
        case 1: k<1, 0><<<512, 256>>>(iters, rnd, out, cyc); break;
        case 2: k<0, 8><<<512, 256>>>(iters, rnd, out, cyc); break;
        case 3: k<1, 8><<<512, 256>>>(iters, rnd, out, cyc); break;
        case 4: k<0, 16><<<512, 256>>>(iters, rnd, out, cyc); break;
        default: k<1, 16><<<512, 256>>>(iters, rnd, out, cyc); break;
      }
      if (rep == 0 && shape == 0) printf("-- sleep %d x 64 cycles per tile\n", sl * 8);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      const double waves = 2048.0, ops = (double)iters * 128 * 32 * 128 * 2;
      printf("shape %s: %.3f ms, %.0f cycles per 128x32 tile per wave, %.0f TOP/s, clock %.2f GHz\n",
             shape ? "16x16x64" : "32x32x32", ms, (double)c / waves / iters, waves * ops / (ms * 1e-3) / 1e12,
             (double)c / waves / (ms * 1e-3) / 1e9);
    }
  return 0;
}
