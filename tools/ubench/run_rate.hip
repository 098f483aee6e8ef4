// Microbenchmark: the rate at which SHORT RUNS come out of HBM -- the access pattern of hs_join8r_kernel
// (per work item one run of 128 x 16-byte packed members and one of 128 x 4-byte records, at increasing
// addresses with gaps between them: the probed buckets of a bucket-ordered table).  Nothing is computed: every
// wave takes items in turn, loads the item's bytes (16 bytes per lane and load, as the kernel does), two items
// ahead, and xors them into a register.  Parameters: run length, a second (short) run per item or not, the gap
// between runs (= density of the probed buckets), waves per CU.
//   hipcc --offload-arch=gfx950 -O3 -o run_rate run_rate.hip && ./run_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// item i: RUN bytes at a + i * stride_a (RUN = 256 * LOADS: 64 lanes x 16 bytes per load -- of which the
// kernel's lanes read 3/4 twice over; here every lane its own 16 bytes, fewer loads for the same bytes),
// and -- SECOND -- 512 bytes at b + i * stride_b (lanes 0..31 x 16 bytes)
template <int LOADS, bool SECOND, int DEPTH>
__global__ __launch_bounds__(256) void k(const char* __restrict__ a, const char* __restrict__ b, uint64_t stride_a,
                                         uint64_t stride_b, uint32_t n_items, uint32_t* counter, uint32_t G,
                                         uint32_t* out) {
  const int lane = threadIdx.x & 63;
  uint4 buf[DEPTH][LOADS + (SECOND ? 1 : 0)];
  uint32_t keep = 0;
  uint32_t item = 0, chunk_end = 0;
  auto next_item = [&]() -> uint32_t {
    if (item == chunk_end) {
      uint32_t v = 0;
      if (lane == 0) v = atomicAdd(counter, G);
      item = __builtin_amdgcn_readfirstlane(v);
      chunk_end = item + G;
    }
    return item++;
  };
  auto load = [&](uint4 (&r)[LOADS + (SECOND ? 1 : 0)], uint32_t it) {
    const char* pa = a + (uint64_t)it * stride_a + 16 * lane;
#pragma unroll
    for (int t = 0; t < LOADS; ++t) r[t] = *reinterpret_cast<const uint4*>(pa + 1024 * t);
    if (SECOND) r[LOADS] = *reinterpret_cast<const uint4*>(b + (uint64_t)it * stride_b + 16 * (lane & 31));
  };
  uint32_t its[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    its[d] = next_item();
    load(buf[d], its[d] < n_items ? its[d] : 0);
  }
  bool more = true;
  while (more) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (more) {
        if (its[d] >= n_items) {
          more = false;
        } else {
#pragma unroll
          for (int t = 0; t < LOADS + (SECOND ? 1 : 0); ++t) keep ^= buf[d][t].x ^ buf[d][t].y ^ buf[d][t].z ^ buf[d][t].w;
          its[d] = next_item();
          load(buf[d], its[d] < n_items ? its[d] : 0);
        }
      }
    }
  }
  if (keep == 0xdeadbeefu) out[threadIdx.x] = keep;
}

int main(int argc, char** argv) {
  const size_t bytes_a = (size_t)(argc > 1 ? atof(argv[1]) : 96.0) * (1ull << 30);  // the "packed" array
  const size_t bytes_b = bytes_a / 4;                                               // the "record" array
  char *a, *b;
  uint32_t *cnt, *out;
  CK(hipMalloc(&a, bytes_a + (1 << 20)));
  CK(hipMalloc(&b, bytes_b + (1 << 20)));
  CK(hipMalloc(&cnt, 4));
  CK(hipMalloc(&out, 4096));
  CK(hipMemset(a, 1, bytes_a));
  CK(hipMemset(b, 2, bytes_b));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int dev_cus = 256;
#define RUN(LOADS, SECOND, DEPTH, WPC, DENS)                                                                   \
  {                                                                                                            \
    /* density DENS = probed share of the table: stride = run / DENS */                                        \
    const uint64_t run_a = 1024ull * LOADS, stride_a = (uint64_t)(run_a / (DENS)) & ~15ull;                    \
    const uint64_t stride_b = stride_a / 4 & ~15ull;                                                           \
    const uint32_t n_items = (uint32_t)(bytes_a / stride_a) - 1;                                               \
    const int blocks = dev_cus * (WPC) / 4;                                                                    \
    float best = 1e30f;                                                                                        \
    for (int rep = 0; rep < 3; ++rep) {                                                                        \
      CK(hipMemset(cnt, 0, 4));                                                                                \
      CK(hipEventRecord(e0));                                                                                  \
      k<LOADS, SECOND, DEPTH><<<blocks, 256>>>(a, b, stride_a, stride_b, n_items, cnt, 256, out);               \
      CK(hipEventRecord(e1));                                                                                  \
      CK(hipEventSynchronize(e1));                                                                             \
      float ms;                                                                                                \
      CK(hipEventElapsedTime(&ms, e0, e1));                                                                    \
      best = ms < best ? ms : best;                                                                            \
    }                                                                                                          \
    const double by = (double)n_items * (run_a + ((SECOND) ? 512.0 : 0.0));                                    \
    printf("run %5llu B%s  depth %d  %2d waves/CU  density %.3f  items %9u  %8.3f ms  %6.2f TB/s\n",            \
           (unsigned long long)run_a, (SECOND) ? " + 512 B" : "        ", DEPTH, WPC, (double)(DENS), n_items,  \
           best, by / best * 1e-9);                                                                            \
  }
  // hs_join8r_kernel's shape: 2 KB + 512 B per item, two items ahead, 8 waves per CU
  RUN(2, true, 2, 8, 0.03)
  RUN(2, true, 2, 8, 0.25)
  RUN(2, true, 2, 8, 1.0)
  RUN(2, false, 2, 8, 0.03)
  RUN(2, false, 2, 8, 0.25)
  RUN(2, false, 2, 8, 1.0)
  // more in flight
  RUN(2, true, 4, 8, 0.03)
  RUN(2, true, 2, 16, 0.03)
  RUN(2, true, 4, 16, 0.03)
  RUN(2, true, 4, 16, 0.25)
  RUN(2, false, 4, 16, 0.03)
  // other run lengths, plenty in flight
  RUN(1, false, 4, 16, 0.03)
  RUN(4, false, 4, 16, 0.03)
  RUN(8, false, 2, 16, 0.03)
  RUN(8, false, 2, 16, 1.0)
  return 0;
}
