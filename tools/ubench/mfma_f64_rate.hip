// Microbenchmark: v_mfma_f64_16x16x4_f64 issue rate (NACC independent accumulators per wave) and
// its C/D layout, against the v_mul_f64 + v_add_f64 pair the exact hash kernel uses.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef double doublex4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void kmfma(int iters, double* out) {
  doublex4 acc[NACC];
  for (int t = 0; t < NACC; ++t) acc[t] = doublex4{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    a += 1e-9;
  }
  double s = 0;
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void kvalu(int iters, double* out) {
  double acc[16];
  for (int t = 0; t < 16; ++t) acc[t] = 0;
  double a = 1.0 + threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = __dadd_rn(acc[t], __dmul_rn(a, 1.0 + t));
    a += 1e-9;
  }
  double s = 0;
  for (int t = 0; t < 16; ++t) s += acc[t];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// layout probe: A[m][k] = m + 100 k, B[k][n] = (k == kk) ? n + 1 : 0 -> D[m][n] = (m + 100 kk)(n + 1)
__global__ void klayout(double* out) {
  const int l = threadIdx.x;
  const int m = l & 15, k = l >> 4;
  const double a = m + 100.0 * k;
  const double b = (k == 2) ? (double)((l & 15) + 1) : 0.0;
  doublex4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

int main() {
  double* out;
  hipMalloc(&out, 8 * 256 * 2048);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  float ms;
#define RUN(NAME, LAUNCH, FLOPS)                                               \
  for (int rep = 0; rep < 2; ++rep) {                                          \
    hipEventRecord(e0);                                                        \
    LAUNCH;                                                                    \
    hipEventRecord(e1);                                                        \
    hipEventSynchronize(e1);                                                   \
    hipEventElapsedTime(&ms, e0, e1);                                          \
    if (rep) printf("%-28s %8.3f ms  %7.2f TFLOP/s\n", NAME, ms, (FLOPS) / ms / 1e9); \
  }
  for (int bpc = 1; bpc <= 2; ++bpc) {
    const double waves = 256.0 * bpc * 4;
    char nm[64];
    snprintf(nm, sizeof nm, "mfma f64 x4 acc, %d blk/CU", bpc);
    RUN(nm, (kmfma<4><<<256 * bpc, 256>>>(iters, out)), waves * iters * 4 * 2048.0);
    snprintf(nm, sizeof nm, "mfma f64 x8 acc, %d blk/CU", bpc);
    RUN(nm, (kmfma<8><<<256 * bpc, 256>>>(iters, out)), waves * iters * 8 * 2048.0);
    snprintf(nm, sizeof nm, "valu mul+add f64, %d blk/CU", bpc);
    RUN(nm, (kvalu<<<256 * bpc, 256>>>(iters, out)), waves * iters * 16 * 64 * 2.0);
  }
  klayout<<<1, 64>>>(out);
  double h[256];
  hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  // expected D[m][n] = (m + 200)(n + 1); report which (m, n) lane l / reg r holds
  int ok_guide = 1;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int n = l & 15, m = (l >> 4) + 4 * r;
      if (h[l * 4 + r] != (m + 200.0) * (n + 1)) ok_guide = 0;
    }
  printf("layout col=lane&15, row=(lane>>4)+4*reg: %s\n", ok_guide ? "yes" : "NO");
  if (!ok_guide)
    for (int l = 0; l < 64; l += 5) printf("  lane %d: %g %g %g %g\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
