// Microbenchmark: issue rate of v_mfma_i32_32x32x32_i8 per SIMD, alone and with a second resident
// wave, with and without a VALU epilogue (32 bitop3) per 16 MFMAs.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx16 __attribute__((ext_vector_type(16)));

template <int EPI>
__global__ __launch_bounds__(256, 2) void k(int iters, int* out, unsigned long long* cyc) {
  intx4 A[4][4], B[4];
  for (int t = 0; t < 4; ++t)
    for (int s = 0; s < 4; ++s) A[t][s] = intx4{(int)threadIdx.x + t, s, t * s, 1};
  for (int s = 0; s < 4; ++s) B[s] = intx4{(int)threadIdx.x, s, 2, 3};
  uint32_t keep = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    intx16 acc[4];
    for (int t = 0; t < 4; ++t)
      for (int i = 0; i < 16; ++i) acc[t][i] = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[t][s], B[s], acc[t], 0, 0, 0);
    if (EPI == 1) {
      uint32_t sall = 0xffffffffu;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) sall &= (uint32_t)acc[t][i];
      if (__ballot((int)sall >= 0) == 0x123456789ull) keep += sall;  // never true in practice
      B[0][0] += (int)(sall >> 31);  // loop-carried dependence: no hoisting
    } else {
      B[0][0] += acc[0][0] & 1;
      keep += (uint32_t)(acc[1][1] ^ acc[2][2] ^ acc[3][3]);
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (keep == 0xdeadbeef) out[threadIdx.x] = (int)keep;
  if ((threadIdx.x & 63) == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));
}

int main() {
  int* out;
  unsigned long long* cyc;
  hipMalloc(&out, 4096);
  hipMalloc(&cyc, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int epi = 0; epi < 2; ++epi)
    for (int bpc = 1; bpc <= 2; ++bpc) {
      for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cyc, 0, 8);
        hipEventRecord(e0);
        if (epi) k<1><<<256 * bpc, 256>>>(iters, out, cyc);
        else k<0><<<256 * bpc, 256>>>(iters, out, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double waves = 256.0 * bpc * 4;
        const double mf = (double)iters * 16;
        printf("epi=%d waves/SIMD=%d: %.3f ms, %.1f cycles per MFMA per wave, %.1f TOP/s, clock %.2f GHz\n", epi,
               bpc, ms, (double)c / waves / mf, waves * mf * 65536.0 / (ms * 1e-3) / 1e12,
               (double)c / waves / (ms * 1e-3) / 1e9);
      }
    }
  return 0;
}
