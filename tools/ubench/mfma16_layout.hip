// Checks the operand layout assumed for v_mfma_i32_16x16x64_i8:
//   A (16 x 64): lane l holds row l & 15, K bytes [16 (l >> 4), +16)
//   B (64 x 16): lane l holds column l & 15, K bytes [16 (l >> 4), +16)
//   D (16 x 16): lane l holds column l & 15, rows 4 (l >> 4) + i, i = 0..3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int intx4 __attribute__((ext_vector_type(4)));
__global__ void k(const int8_t* A, const int8_t* B, int* D) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  const intx4 a = *reinterpret_cast<const intx4*>(A + r * 64 + 16 * g);
  const intx4 b = *reinterpret_cast<const intx4*>(B + r * 64 + 16 * g);
  intx4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];
}
int main() {
  int8_t hA[16 * 64], hB[16 * 64]; int hD[256];
  srand(7);
  for (int i = 0; i < 1024; ++i) { hA[i] = (int8_t)(rand() % 255 - 127); hB[i] = (int8_t)(rand() % 255 - 127); }
  int8_t *dA, *dB; int* dD;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
  hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      int ref = 0;
      for (int kk = 0; kk < 64; ++kk) ref += (int)hA[m * 64 + kk] * (int)hB[n * 64 + kk];
      if (ref != hD[m * 16 + n]) ++bad;
    }
  printf("mismatches: %d of 256\n", bad);
  return bad != 0;
}
