// Which XCD (HW_REG_XCC_ID) workgroup b of a launch lands on: round-robin b % 8 on this part.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  const unsigned x = __builtin_amdgcn_s_getreg(6164);  // hwreg(HW_REG_XCC_ID, 0, 4)
  if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
  unsigned* d; hipMalloc(&d, 4 * 64);
  k<<<64, 64>>>(d);
  unsigned h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; ++i) printf("%u%c", h[i], i % 16 == 15 ? '\n' : ' ');
  return 0;
}
