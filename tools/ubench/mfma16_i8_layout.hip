// Checks the operand / result layout of v_mfma_i32_16x16x64_i8 assumed by hs_join8x_kernel:
//   A: lane l holds row (l & 15), bytes k = 16 (l >> 4) .. + 15;  B: lane l holds column (l & 15), same k;
//   D: lane l, register i = element (row 4 (l >> 4) + i, column l & 15).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int intx4 __attribute__((ext_vector_type(4)));
__global__ void k(const int8_t* A, const int8_t* B, int* D) {  // A[16][64], B[16 cols][64], D[16][16]
  const int l = threadIdx.x, r = l & 15, q = l >> 4;
  intx4 a, b, c = {0, 0, 0, 0};
  a = *reinterpret_cast<const intx4*>(A + r * 64 + 16 * q);
  b = *reinterpret_cast<const intx4*>(B + r * 64 + 16 * q);
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(4 * q + i) * 16 + r] = c[i];
}
int main() {
  int8_t hA[16 * 64], hB[16 * 64];
  srand(1);
  for (int i = 0; i < 16 * 64; ++i) { hA[i] = (int8_t)(rand() % 255 - 127); hB[i] = (int8_t)(rand() % 255 - 127); }
  int8_t *dA, *dB; int* dD; int hD[256];
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      int s = 0;
      for (int kk = 0; kk < 64; ++kk) s += (int)hA[m * 64 + kk] * (int)hB[n * 64 + kk];
      if (s != hD[m * 16 + n]) ++bad;
    }
  printf("v_mfma_i32_16x16x64_i8 layout: %s (%d mismatches)\n", bad ? "WRONG" : "as assumed", bad);
  return bad != 0;
}
