// Microbenchmark: 16 int8 MFMAs + sign-test epilogue per 4 KB "query tile" fetched from global
// memory straight into registers, one tile ahead (the inner loop of hs_join8w_kernel), with
// different address patterns:  0 no loads; 1 every wave walks the SAME 256 KB window in the same
// order; 2 same window, block-dependent starting phase; 3 a private 256 KB window per block.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void loadb(intx4 (&B)[4], const uint4* base, int lane) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint4 v = base[s * 64 + lane];
    B[s] = intx4{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(int iters, const uint4* __restrict__ buf, int tiles, int* out,
                                            unsigned long long* cyc) {
  const int lane = threadIdx.x & 63;
  intx4 A[4][4], Ba[4], Bb[4];
  for (int t = 0; t < 4; ++t)
    for (int s = 0; s < 4; ++s) A[t][s] = intx4{(int)threadIdx.x + t, s, t * s, 1};
  for (int s = 0; s < 4; ++s) Ba[s] = Bb[s] = intx4{(int)threadIdx.x, s, 2, 3};
  const uint4* win = buf + (MODE == 3 ? (size_t)blockIdx.x * tiles * 256 : 0);
  uint32_t tile = MODE == 2 ? (blockIdx.x * 7u) % (uint32_t)tiles : 0u;
  if (MODE) {
    loadb(Ba, win + (size_t)tile * 256, lane);
    tile = (tile + 1) % tiles;
    loadb(Bb, win + (size_t)tile * 256, lane);
    tile = (tile + 1) % tiles;
  }
  uint32_t keep = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      intx4 (&B)[4] = half ? Bb : Ba;
      intx16 acc[4];
      for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[t][s], B[s], acc[t], 0, 0, 0);
      if (MODE) {
        loadb(B, win + (size_t)tile * 256, lane);
        tile = tile + 1 == (uint32_t)tiles ? 0 : tile + 1;
      }
      uint32_t sall = 0xffffffffu;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) sall &= (uint32_t)acc[t][i];
      if (__ballot((int)sall >= 0) == 0x123456789ull) keep += sall;
      if (!MODE) B[0][0] += (int)(sall >> 31);
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (keep == 0xdeadbeef) out[threadIdx.x] = (int)keep;
  if (lane == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));
}

int main() {
  int* out;
  unsigned long long* cyc;
  uint4* buf;
  const int tiles = 64;  // 64 x 4 KB = 256 KB window
  const int blocks = 512;
  CK(hipMalloc(&out, 4096));
  CK(hipMalloc(&cyc, 8));
  CK(hipMalloc(&buf, (size_t)blocks * tiles * 4096));
  CK(hipMemset(buf, 0x81, (size_t)blocks * tiles * 4096));  // negative bytes: products positive, never mind
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int mode = 0; mode < 4; ++mode)
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(cyc, 0, 8));
      CK(hipEventRecord(e0));
      switch (mode) {
        case 0: k<0><<<blocks, 256>>>(iters, buf, tiles, out, cyc); break;
        case 1: k<1><<<blocks, 256>>>(iters, buf, tiles, out, cyc); break;
        case 2: k<2><<<blocks, 256>>>(iters, buf, tiles, out, cyc); break;
        case 3: k<3><<<blocks, 256>>>(iters, buf, tiles, out, cyc); break;
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long c;
      CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      const double waves = blocks * 4.0, mf = (double)iters * 16;
      printf("mode %d: %.3f ms, %.0f cycles per tile per wave, %.0f TOP/s, B stream %.2f TB/s, clock %.2f GHz\n",
             mode, ms, (double)c / waves / iters, waves * mf * 65536.0 / (ms * 1e-3) / 1e12,
             mode ? waves * iters * 4096.0 / (ms * 1e-3) / 1e12 : 0.0, (double)c / waves / (ms * 1e-3) / 1e9);
    }
  return 0;
}
